#!/bin/bash
# A/B builds of the kernel files: scripts/ab_build.sh NAME "-DFOO=1 ..." -> gnn_computing_amd/csrc/build/ab/libgnnagg_NAME.so
# (select at run time with GNNAGG_LIB=<path>).
set -e
cd "$(dirname "$0")/../gnn_computing_amd/csrc"
make -s -j4
mkdir -p build/ab
objs=""
for f in agg_gcn agg_gat agg_span aux_kernels plan_gpu; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form $2 -c $f.hip -o build/ab/${f}_$1.o &
  objs="$objs build/ab/${f}_$1.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libgnnagg_$1.so $objs build/api.o build/api_flat.o build/api_extras.o build/host_graph.o build/reorder.o build/dist_rccl.o -lgomp -ldl -Wl,--exclude-libs,ALL
echo build/ab/libgnnagg_$1.so
