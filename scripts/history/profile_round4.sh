#!/bin/bash
# Round-4 evidence in one pass on the GPU box (through gpurun):  scripts/profile_round4.sh gpurun_out/r4/final
# Everything lands under $OUT; the files to keep are copied into profiles/r04/ afterwards (scripts/collect_r04.sh).
set -u
OUT=${1:-gpurun_out/r4/final}
mkdir -p $OUT
export TMPDIR=/tmp
export ROUND=r04
# 1. bench under rocprofv3, PMC passes per config (traffic, L2 hit, DRAM share of the fabric reads), pmc_traffic.json, bench lines
STEPS=${STEPS:-50} WARM=${WARM:-5} SQ=${SQ:-1} bash scripts/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
# 2. the FETCH_SIZE / WRITE_SIZE calibration on known byte counts
bash scripts/fetch_calibration.sh $OUT/cal > $OUT/fetch_calibration.log 2>&1
cp $OUT/cal/fetch_calibration.txt $OUT/cal/fetch_calibration.json $OUT/ 2>/dev/null
rm -rf $OUT/cal/cal_pmc_* $OUT/cal/cal_trace
# 3. plan construction (device builder against the host builders), rows modes, GAT chains
python3 scripts/exp_plan_time.py reddit > $OUT/plan_time.txt 2>&1
ROWS_MEDIUM=-1 python3 scripts/exp_rows_mode.py A P1 R G > $OUT/rows_mode.txt 2>&1   # rows_medium_-1_us: without the medium class
[ -x scripts/micro/bin/fma_chain ] && scripts/micro/bin/fma_chain > $OUT/fma_chain.txt 2>&1
python3 scripts/exp_rows_blocked_gat.py > $OUT/rows_gat.txt 2>&1
# 4. the N > 1 bench line on this one GPU (two ranks over gloo: test hooks), launched the way the driver launches N = 1
BENCH_ONE_GPU=1 BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err
# 4b. the same line on its default transport, the one-call C-ABI step, with the nccl entry points served by the test double
BENCH_ONE_GPU=1 BENCH_BACKEND=gloo BENCH_TRANSPORT=rccl BENCH_NO_FALLBACK=1 BENCH_PRODUCTS=0 GNNAGG_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_cabi_step_double.json 2> $OUT/bench_2ranks_cabi_step_double.err
# 5. drivers, the reference's kernels beside this library, GEMM, 3-layer forward
python3 scripts/run_drivers.py 128 > $OUT/drivers.txt 2>&1
python3 tests/perf_reference_on_mi355x.py > $OUT/reference_on_mi355x.jsonl 2> $OUT/reference_on_mi355x.err
python3 scripts/bench_gemm.py > $OUT/gemm_final.txt 2>&1
python3 examples/forward_3layer.py --model our_GCN --dataset arxiv > $OUT/forward3_gcn.txt 2>&1
ls -la $OUT
