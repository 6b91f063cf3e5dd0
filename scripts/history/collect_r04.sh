#!/bin/bash
# Copies what scripts/profile_round4.sh left under gpurun_out/r4/final into profiles/r04/ (the tracked evidence).
set -u
SRC=${1:-gpurun_out/r4/final}
DST=profiles/r04
mkdir -p $DST
for f in bench.json bench_arms.txt bench_configs.jsonl bench_kernel_stats.csv pmc_traffic.json drivers.txt fetch_calibration.txt fetch_calibration.json \
         plan_time.txt rows_mode.txt rows_gat.txt fma_chain.txt bench_2ranks_one_gpu.json bench_2ranks_cabi_step_double.json reference_on_mi355x.jsonl gemm_final.txt forward3_gcn.txt; do
  [ -s $SRC/$f ] && grep -v "amdgpu.ids" $SRC/$f > $DST/$f
done
for f in $SRC/summary_*.txt; do [ -s $f ] && cp $f $DST/; done
ls -la $DST
