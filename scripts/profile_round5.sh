#!/bin/bash
# Round-5 evidence in one pass on the GPU box (through gpurun):  scripts/profile_round5.sh gpurun_out/r5/final
# Everything lands under $OUT; scripts/collect_r05.sh copies the files to keep into profiles/r05/.
set -u
OUT=${1:-gpurun_out/r5/final}
mkdir -p $OUT
export TMPDIR=/tmp
export ROUND=r05
FAKE=$PWD/tests/fake_rccl/libfakerccl.so
# 1. the driver's command under rocprofv3 (kernel trace + stats), PMC passes per config (A, A_rows, R, G, P1: traffic, L2 hit), pmc_traffic.json,
#    then the bench line itself (configs sub-records included) and the per-config lines
CONFIGS="A_rows R G P1" STEPS=${STEPS:-20} WARM=${WARM:-5} SQ=${SQ:-0} bash scripts/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
# 2. the driver's exact command once more, as the driver runs it (20 steps, 5 warm-up, cpu_baseline included)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
# 3. N > 1 lines on this ONE GPU -- functional checks of the row-partitioned step, labelled as such by the line itself
BENCH_ONE_GPU=1 BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err
D="BENCH_ONE_GPU=1 BENCH_BACKEND=gloo BENCH_TRANSPORT=rccl BENCH_NO_FALLBACK=1 GNNAGG_RCCL_LIB=$FAKE"
env $D BENCH_PRODUCTS=0 python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_cabi_step_double.json 2> $OUT/bench_2ranks_cabi_step_double.err
env $D BENCH_PRODUCTS=0 BENCH_STAGES=auto python3 bench.py --gpus 8 --steps 5 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_auto.json 2> $OUT/bench_8ranks_double_auto.err
env $D BENCH_PRODUCTS=0 BENCH_STAGES=owner python3 bench.py --gpus 8 --steps 5 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_owner.json 2> $OUT/bench_8ranks_double_owner.err
env $D BENCH_STAGES=auto python3 bench.py --gpus 8 --config P --steps 3 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_P.json 2> $OUT/bench_8ranks_double_P.err
# 4. drivers (the reference's figures on this library), the reference's kernels beside it, GEMM, 3-layer forward
python3 scripts/run_drivers.py 128 > $OUT/drivers.txt 2>&1
python3 tests/perf_reference_on_mi355x.py > $OUT/reference_on_mi355x.jsonl 2> $OUT/reference_on_mi355x.err
python3 scripts/bench_gemm.py > $OUT/gemm_final.txt 2>&1
python3 examples/forward_3layer.py --model our_GCN --dataset arxiv > $OUT/forward3_gcn.txt 2>&1
# 5. second tier of the GPU suite: the 600-case fuzz against the reference's kernels and the 8-rank spawn case
REF_FUZZ=600 GNNAGG_TEST_TIER=2 python3 -m pytest tests/test_gpu_reference.py::test_reference_fuzz "tests/test_gpu_dist.py::test_cabi_step_with_several_peers_on_one_gpu" -q > $OUT/second_tier.txt 2>&1
ls -la $OUT
