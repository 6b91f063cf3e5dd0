#!/bin/bash
# MFMA / wait counters of the wide GEMM kernel (k_dense_nn_big) on the 512 -> 128 layer.  usage: exp_gemm_pmc.sh <out dir>
OUT=${1:-gpurun_out/r3/gemm_pmc}
mkdir -p $OUT
export GEMM_SHAPES=${GEMM_SHAPES:-169343x512x128}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o t -- python3 scripts/bench_gemm.py > $OUT/log_$tag.txt 2>&1
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dense_nn" in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-62s %-32s n=%3d avg=%.4g" % (k, c, n, s / n))
PY
