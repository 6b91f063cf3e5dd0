#!/bin/bash
# Round-6 evidence in one pass on the GPU box (through gpurun):  scripts/profile_round6.sh gpurun_out/r6/final
# Everything lands under $OUT; scripts/collect_r06.sh copies the files to keep into profiles/r06/.
set -u
OUT=${1:-gpurun_out/r6/final}
mkdir -p $OUT
export TMPDIR=/tmp
export ROUND=r06
FAKE=$PWD/tests/fake_rccl/libfakerccl.so
# 0. the generator's permutation of the products-shaped graph is made once (38 s) and found again by every later process of this call
python3 -c "import bench, numpy, torch, gnn_computing_amd as g; bench.np = numpy; p, i = g.graph.dataset('products', device='cuda'); print(bench.load_with_locality_reorder('products', p.cpu().numpy(), i.cpu().numpy())[3:])" > $OUT/reorder_cache.txt 2>&1
# 1. the driver's command under rocprofv3 (kernel trace + stats), PMC passes per config (A, A_rows, R, G, P1, P1_reorder: traffic, L2 hit), pmc_traffic.json,
#    then the bench line itself (configs sub-records included) and the per-config lines
CONFIGS="A_rows R G P1 P1_reorder" STEPS=${STEPS:-20} WARM=${WARM:-5} SQ=${SQ:-0} bash scripts/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
# 2. the driver's exact command once more, as the driver runs it (20 steps, 5 warm-up, cpu_baseline included)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
# 3. N > 1 lines on this ONE GPU -- functional checks of the row-partitioned step over the ASYNCHRONOUS test double, labelled as such by the line itself
BENCH_ONE_GPU=1 BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err
D="BENCH_ONE_GPU=1 BENCH_BACKEND=gloo BENCH_TRANSPORT=rccl BENCH_NO_FALLBACK=1 GNNAGG_RCCL_LIB=$FAKE"
env $D BENCH_PRODUCTS=0 python3 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu > $OUT/bench_2ranks_cabi_step_double.json 2> $OUT/bench_2ranks_cabi_step_double.err
env $D BENCH_PRODUCTS=0 python3 bench.py --gpus 8 --steps 5 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_auto.json 2> $OUT/bench_8ranks_double_auto.err
env $D BENCH_PRODUCTS=0 BENCH_STAGES=owner python3 bench.py --gpus 8 --steps 5 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_owner.json 2> $OUT/bench_8ranks_double_owner.err
env $D python3 bench.py --gpus 8 --config P --steps 3 --warmup 1 --no-cpu > $OUT/bench_8ranks_cabi_step_double_P.json 2> $OUT/bench_8ranks_double_P.err
# 4. the double itself, the null-stream finding, drivers, the reference's kernels beside this library, GEMM, 3-layer forward
for w in 2 4; do rm -f /tmp/fakerccl_idf$w; for r in $(seq 0 $((w-1))); do RANK=$r WORLD_SIZE=$w timeout 120 tests/fake_rccl/selftest.out --idfile /tmp/fakerccl_idf$w --rounds 20 >> $OUT/fake_rccl_selftest.jsonl 2>> $OUT/fake_rccl_selftest.err & done; wait; done
python3 tests/perf_reorder_discrepancy.py host > $OUT/null_stream.jsonl 2> $OUT/null_stream.err
python3 tests/perf_reorder_discrepancy.py xcd >> $OUT/null_stream.jsonl 2>> $OUT/null_stream.err
python3 scripts/run_drivers.py 128 > $OUT/drivers.txt 2>&1
python3 tests/perf_reference_on_mi355x.py > $OUT/reference_on_mi355x.jsonl 2> $OUT/reference_on_mi355x.err
python3 scripts/bench_gemm.py > $OUT/gemm_final.txt 2>&1
python3 examples/forward_3layer.py --model our_GCN --dataset arxiv > $OUT/forward3_gcn.txt 2>&1
# 5. second tier of the GPU suite: the older kernel forms and the backward entry points on libgnnagg_extras.so, the 600-case fuzz against the
#    reference's kernels, the plain several-peer cases and the 8-rank spawn
GNNAGG_LIB=$PWD/gnn_computing_amd/libgnnagg_extras.so GNNAGG_TEST_TIER=2 python3 -m pytest tests/test_gpu_blocked.py tests/test_gpu_parity.py tests/test_cabi.py -q > $OUT/second_tier_extras.txt 2>&1
REF_FUZZ=600 GNNAGG_TEST_TIER=2 python3 -m pytest tests/test_gpu_reference.py::test_reference_fuzz tests/test_gpu_dist.py tests/test_gpu_bench_contract.py::test_failed_nccl_backend_falls_back_to_gloo_in_fresh_processes -q > $OUT/second_tier.txt 2>&1
GNNAGG_TEST_STREAM=side python3 -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_bench_contract.py > $OUT/second_tier_side_stream.txt 2>&1
ls -la $OUT
