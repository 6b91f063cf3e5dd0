#!/bin/bash
# Copies what scripts/profile_round3.sh left under gpurun_out/r3/final into profiles/r03/ (the tracked evidence).
set -u
SRC=${1:-gpurun_out/r3/final}
DST=profiles/r03
mkdir -p $DST
for f in bench.json bench_arms.txt bench_configs.jsonl bench_kernel_stats.csv pmc_traffic.json drivers.txt drivers_f32.txt fig9_ref_kernels.txt \
         fig9_reorder_l2.txt p1_reorder.txt reference_on_mi355x.jsonl rows_mode.txt forward3_gcn.txt; do
  [ -s $SRC/$f ] && cp $SRC/$f $DST/$f
done
for f in $SRC/summary_*.txt; do [ -s $f ] && cp $f $DST/; done
ls -la $DST
