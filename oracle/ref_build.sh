#!/bin/bash
# oracle/ref_build.sh -- builds oracle/_ref/libref_gnn.so: the REFERENCE's own aggregation path, compiled for gfx950.
#
# The reference is CUDA (nvcc + cuSPARSE / cuBLAS / cuRAND headers, include/util.h:4-8); this image has no CUDA toolkit but
# it has ROCm's own translator and the ROCm counterparts of those libraries.  So the reference's sources are translated
# WHERE THEY LIE with hipify-perl into a scratch directory outside the repo, compiled together with oracle/ref_shim.hip
# (this repo's C entry points over the reference's classes) by hipcc against hipSPARSE / hipBLAS / hipRAND, and the scratch
# directory is removed.  Nothing of the reference is written into the repo; the only output is oracle/_ref/libref_gnn.so
# (git-ignored, travels to the GPU box like the other built libraries).
#
# Two textual substitutions are applied to the translated copies, both forced by the hardware and neither in the arithmetic
# of the anchored kernels:
#   1. `0xffffffff` shuffle masks -> 64-bit all-ones (aggregator.h:5-6 defines __shfl as __shfl_sync(0xffffffff, ...); HIP's
#      *_sync shuffles take a 64-bit mask of the 64-lane wavefront).  The shuffles of the anchored kernels keep their explicit
#      32-lane width argument.
#   2. the PTX special-register reads of the Figure-8 clock helpers (aggr_gcn.h:143-156: %globaltimer, %smid) ->
#      wall_clock64() / __smid().  Instrumentation only; the shim never calls the clock kernels.
# Two compiler flags on top of the reference's own (-O2):
#   -include cstring      src/data.cu:96-98 calls strlen without including <cstring> (nvcc's headers pull it in)
#   -Xarch_device -DNDEBUG   HIP's *_sync shuffles assert (trap) unless every lane named in the mask is active.  The
#      reference's 32-lane warps share a 64-lane wavefront with their neighbour and legitimately diverge from it (rows of
#      different length), so the all-ones mask names inactive lanes.  Device pass only: the host-side asserts stay.
# No GPU is needed to build.  Skips quietly (exit 0) when the reference tree or hipify-perl is missing: oracle/_ref is optional.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROCM=${ROCM_PATH:-/opt/rocm}
HIPIFY=$ROCM/bin/hipify-perl
if [ ! -d "$REF/include" ] || [ ! -x "$HIPIFY" ]; then
  echo "oracle/_ref: no reference tree at $REF (or no hipify-perl): not built"
  exit 0
fi
OUT="$HERE/_ref/libref_gnn.so"
if [ -f "$OUT" ] && [ -z "$FORCE" ]; then   # up to date?  (the translation takes ~100 s)
  newer=$(find "$REF/include" "$REF/src" "$HERE/ref_shim.hip" "$HERE/ref_build.sh" -newer "$OUT" -type f 2>/dev/null | head -1)
  if [ -z "$newer" ]; then echo "oracle/_ref/libref_gnn.so is up to date"; exit 0; fi
fi
GEN=$(mktemp -d "${TMPDIR:-/tmp}/gnnref.XXXXXX")
trap 'rm -rf "$GEN"' EXIT
mkdir -p "$GEN/include" "$GEN/src" "$HERE/_ref"
# (args.hxx and dbg.h are vendored host-only libraries: used from the reference tree as they are, see -I"$REF/include" below)
# (hipify-perl takes 2-10 s per file: the translations run side by side)
for f in "$REF"/include/*.h; do [ "$(basename "$f")" = dbg.h ] || "$HIPIFY" "$f" > "$GEN/include/$(basename "$f")" 2>/dev/null & done
for f in data util; do "$HIPIFY" "$REF/src/$f.cu" > "$GEN/src/$f.hip" 2>/dev/null & done
wait
sed -i 's/0xffffffff\b/0xffffffffffffffffULL/g' "$GEN"/include/aggregator.h "$GEN"/include/aggr_gat.h "$GEN"/include/aggr_gcn.h \
    "$GEN"/include/aggr_nn.h "$GEN"/include/aggr_sddmm.h "$GEN"/include/spmm.h "$GEN"/include/sample.h
sed -i -e 's/asm volatile("mov.u64 %0, %%globaltimer;" : "=l"(first_reading));/first_reading = wall_clock64();/' \
       -e 's/asm volatile("mov.u32 %0, %%globaltimer_hi;" : "=r"(second_reading));/second_reading = (uint32_t)(wall_clock64() >> 32);/' \
       -e 's/asm("mov.u32 %0, %smid;" : "=r" (smid));/smid = __smid();/' "$GEN/include/aggr_gcn.h"
# -O2 as the reference's CMakeLists.txt:40; hipcc contracts a * b + c into an fma like nvcc does (the oracle's fmaf chain)
"$ROCM/bin/hipcc" --offload-arch=gfx950 -O2 -std=c++17 -fopenmp -fPIC -shared -fvisibility=hidden -w -DDBG_MACRO_DISABLE -include cstring -Xarch_device -DNDEBUG \
    -I"$GEN/include" -I"$REF/include" -I"$ROCM/include/hipsparse" -I"$ROCM/include/hipblas" -I"$ROCM/include/hiprand" \
    "$HERE/ref_shim.hip" "$GEN/src/data.hip" "$GEN/src/util.hip" -o "$OUT" \
    -L"$ROCM/lib" -lhipsparse -lhipblas -lhiprand -Wl,-rpath,"$ROCM/lib"
echo "oracle/_ref/libref_gnn.so built from $REF"
