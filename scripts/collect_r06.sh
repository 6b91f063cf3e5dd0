#!/bin/bash
# Copies what scripts/profile_round6.sh left under gpurun_out/r6/final into profiles/r06/ (the tracked evidence).
set -u
SRC=${1:-gpurun_out/r6/final}
DST=profiles/r06
mkdir -p $DST
for f in bench.json bench_driver_command.json bench_arms.txt bench_configs.jsonl bench_kernel_stats.csv pmc_traffic.json drivers.txt \
         bench_2ranks_one_gpu.json bench_2ranks_cabi_step_double.json bench_8ranks_cabi_step_double_auto.json bench_8ranks_cabi_step_double_owner.json \
         bench_8ranks_cabi_step_double_P.json reference_on_mi355x.jsonl gemm_final.txt forward3_gcn.txt second_tier.txt second_tier_extras.txt second_tier_side_stream.txt hw_queues.txt \
         fake_rccl_selftest.jsonl null_stream.jsonl; do
  [ -s $SRC/$f ] && grep -v "amdgpu.ids" $SRC/$f > $DST/$f
done
for f in $SRC/summary_*.txt; do [ -s $f ] && cp $f $DST/; done
ls -la $DST
