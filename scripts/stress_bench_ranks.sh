#!/bin/bash
# Repeats the 2-rank functional bench lines (ranks sharing the one GPU) to catch rare first-step oracle mismatches (exit 17):
#   scripts/stress_bench_ranks.sh OUTDIR N      -- per failing run the full stderr is kept in OUTDIR
OUT=${1:-gpurun_out/stress}; N=${2:-15}
mkdir -p $OUT
FAKE=$PWD/tests/fake_rccl/libfakerccl.so
fail=0
for i in $(seq 1 $N); do
  for kind in double gloo; do
    if [ $kind = double ]; then E="BENCH_TRANSPORT=rccl BENCH_NO_FALLBACK=1 GNNAGG_RCCL_LIB=$FAKE"; else E=""; fi
    env BENCH_ONE_GPU=1 BENCH_BACKEND=gloo BENCH_PRODUCTS=0 $E timeout 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu > $OUT/run.json 2> $OUT/run.err
    rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); cp $OUT/run.err $OUT/fail_${kind}_$i.err; echo "run $i $kind: rc $rc"; grep -h "differs from the oracle\|watchdog\|Error\|error" $OUT/run.err | head -5; fi
  done
done
echo "failures: $fail of $((2*N)) runs"
