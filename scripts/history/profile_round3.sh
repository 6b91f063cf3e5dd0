#!/bin/bash
# Round-3 evidence in one pass on the GPU box (through gpurun):  scripts/profile_round3.sh gpurun_out/r3/final
# Everything lands under $OUT; the files to keep are copied into profiles/r03/ afterwards (scripts/collect_r03.sh).
set -u
OUT=${1:-gpurun_out/r3/final}
mkdir -p $OUT
export TMPDIR=/tmp
export ROUND=r03
# 1. bench under rocprofv3, PMC passes per config, pmc_traffic.json, bench lines (A with the fresh counters, R / G / P1)
STEPS=${STEPS:-50} WARM=${WARM:-5} SQ=${SQ:-1} bash scripts/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
# 2. C++ drivers of this repo and the REFERENCE's own drivers on the shim, no environment variables
python3 scripts/run_drivers.py 128 > $OUT/drivers.txt 2>&1
python3 scripts/run_drivers.py 32 > $OUT/drivers_f32.txt 2>&1
# 3. the reference's Figure9/main.cu on the shim under the kernel trace: what its run(.,.,B,0) and run(.,.,B,1) cost by default
python3 - <<PY > $OUT/fig9_ref_setup.log 2>&1
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import gnn_computing_amd as gnc
d = "$OUT/fig9data/"
ptr, idx = gnc.graph.dataset("arxiv")
gnc.graph.write_graph_files(d, "arxiv", ptr.numpy(), idx.numpy(), text=False, dumps=True)
rows, _ = gnc.cluster_reorder(ptr.numpy(), idx.numpy(), order="cache_greedy", cluster_cap=1, cache_rows=8192)
gnc.graph.write_reorder_file(d, "arxiv", np.asarray(rows, np.int32))
PY
for arm in plain reorder; do
  extra=""; [ $arm = reorder ] && extra="--reorder _thres_0.2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fig9ref_$arm -o t -- oracle/_ref/drivers/fig9_ref.out --dataset arxiv \
      --datadir $OUT/fig9data/ --feature-len 128 --nei 32 $extra > $OUT/fig9_ref_$arm.log 2>&1
  python3 - <<PY >> $OUT/fig9_ref_kernels.txt
import csv, glob
f = glob.glob("$OUT/trace_fig9ref_$arm/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gnnagg" in r["Kernel_Name"]]
print("== oracle/_ref/drivers/fig9_ref.out --feature-len 128 --nei 32 ($arm), no environment variables: kernels in launch order")
# 10 x run(x, y, B, 0) then 10 x run(x, y2, B, 1): average the two halves
half = len(rows) // 2
for name, part in (("run(., ., B, 0)", rows[:half]), ("run(., ., B, 1)", rows[half:])):
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in part]
    kn = sorted(set(r["Kernel_Name"][:60] for r in part))
    print("  %-16s %2d launches, avg %.1f us, median %.1f us   %s" % (name, len(us), sum(us) / len(us), sorted(us)[len(us) // 2], kn))
PY
done
rm -rf $OUT/fig9data $OUT/trace_fig9ref_*
# 4. rows mode, reorder generator on the box's cores + P1, the reference's kernels beside this library, Figure-9 L2 evidence, GEMM
python3 scripts/exp_rows_mode.py A P1 R G > $OUT/rows_mode.txt 2>&1
python3 scripts/exp_p1_reorder.py > $OUT/p1_reorder.txt 2>&1
python3 tests/perf_reference_on_mi355x.py > $OUT/reference_on_mi355x.jsonl 2> $OUT/reference_on_mi355x.err
ARMS="none lsh greedy community" bash scripts/profile_fig9.sh $OUT/fig9 > $OUT/fig9_reorder_l2.txt 2>&1
rm -rf $OUT/fig9/trace_* $OUT/fig9/pmc_* $OUT/fig9/fetch_*
python3 scripts/bench_gemm.py > $OUT/gemm.txt 2>&1
python3 examples/forward_3layer.py --model our_GCN --dataset arxiv > $OUT/forward3_gcn.txt 2>&1
ls -la $OUT
