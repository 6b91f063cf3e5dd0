#!/bin/bash
# A/B builds of the kernel file: scripts/ab_build.sh NAME "-DFOO=1 ..." -> gnn_computing_amd/csrc/build/ab/libgnnagg_NAME.so
# (select at run time with GNNAGG_LIB=<path>).
set -e
cd "$(dirname "$0")/../gnn_computing_amd/csrc"
make -s
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form $2 -c kernels.hip -o build/ab/kernels_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libgnnagg_$1.so build/ab/kernels_$1.o build/api.o build/host_graph.o build/reorder.o -lgomp -Wl,--exclude-libs,ALL
echo build/ab/libgnnagg_$1.so
