#!/bin/bash
# drivers/build_reference_drivers.sh -- the drop-in claim, exercised: the REFERENCE's own benchmark drivers
# (/root/reference/Figure8/main.cu, Figure9/main.cu, Figure10/main_a.cu, Figure10/main_b.cu), translated where they lie by ROCm's hipify-perl
# (the mechanical cuda* -> hip* rename, no hand edits) into a scratch directory, compiled against THIS repo's class shim
# (include/compat/: Aggregator_GCN / Aggregator_GAT / load_graph / fullGraph / argParse / matmul_NN with the reference's
# signatures) and linked with libgnnagg.so.  Outputs: oracle/_ref/drivers/fig8_ref.out, fig9_ref.out, fig10a_ref.out, fig10b_ref.out
# (git-ignored).  Figure8/main.cu takes the occupancy of the reference's kernel symbols (aggr_gcn_clock, aggr_gcn_target_clock:
# declared with empty bodies in include/compat/aggr_gcn.h) and sizes its timer buffers from the CUDA launch geometry
# (Figure8/main.cu:76-99): Aggregator_GCN::run_clock takes that geometry as the buffers' capacity.  Its analysis keeps the V100
# constant of :143 (80 SMs) -- drivers/fig8.cpp is the counterpart that knows the MI355X's CU count.
# Skips quietly when the reference tree or hipify-perl is missing.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROCM=${ROCM_PATH:-/opt/rocm}
OUTDIR="$HERE/../oracle/_ref/drivers"   # every reference-derived binary lives under oracle/_ref/ (git-ignored)
if [ ! -d "$REF/Figure9" ] || [ ! -x "$ROCM/bin/hipify-perl" ]; then echo "oracle/_ref/drivers: no reference tree: not built"; exit 0; fi
if [ -f "$OUTDIR/fig10b_ref.out" ] && [ -f "$OUTDIR/fig8_ref.out" ] && [ -z "$FORCE" ]; then   # up to date?
  newer=$(find "$REF/Figure8/main.cu" "$REF/Figure9/main.cu" "$REF/Figure10/main_a.cu" "$REF/Figure10/main_b.cu" "$HERE/../include/compat" "$HERE/../include/gnnagg.h" \
               "$HERE/build_reference_drivers.sh" -newer "$OUTDIR/fig10b_ref.out" -type f 2>/dev/null | head -1)
  if [ -z "$newer" ]; then echo "oracle/_ref/drivers is up to date"; exit 0; fi
fi
GEN=$(mktemp -d "${TMPDIR:-/tmp}/gnnrefdrv.XXXXXX")
trap 'rm -rf "$GEN"' EXIT
mkdir -p "$OUTDIR"
build() {  # <reference source> <output name>
  "$ROCM/bin/hipify-perl" "$REF/$1" > "$GEN/$2.hip" 2>/dev/null
  "$ROCM/bin/hipcc" --offload-arch=gfx950 -O2 -std=c++17 -w -I"$HERE/../include/compat" -I"$HERE/../include" \
      -I"$ROCM/include/hiprand" -I"$ROCM/include/hipblas" "$GEN/$2.hip" -o "$OUTDIR/$2.out" \
      -L"$HERE/../gnn_computing_amd" -lgnnagg -L"$ROCM/lib" -lhiprand -lhipblas -Wl,-rpath,'$ORIGIN/../../../gnn_computing_amd' -Wl,-rpath,"$ROCM/lib"
}
build Figure8/main.cu fig8_ref & p0=$!
build Figure9/main.cu fig9_ref & p1=$!
build Figure10/main_a.cu fig10a_ref & p2=$!
build Figure10/main_b.cu fig10b_ref & p3=$!
wait $p0 && wait $p1 && wait $p2 && wait $p3
echo "oracle/_ref/drivers: the reference's Figure8 / Figure9 / Figure10a / Figure10b drivers built against include/compat + libgnnagg.so"
