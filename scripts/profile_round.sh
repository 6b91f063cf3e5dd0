#!/bin/bash
# Everything profiles/<round>/ holds, in one pass on the GPU box (through gpurun):  scripts/profile_round.sh gpurun_out/r02/final
#   1. rocprofv3 --kernel-trace --stats over the bench command -> bench_kernel_stats.csv + per-arm averages of the dominant kernel
#   2. PMC passes (traffic, L2 hit rate; SQ/TA with SQ=1) over the headline workload and the other configs -> summary_*.txt/json
#   3. pmc_traffic.json (what bench.py reports as roofline.traffic), then the bench lines themselves with it in place
set -u
OUT=${1:-gpurun_out/prof_round}
STEPS=${STEPS:-50}; WARM=${WARM:-5}
mkdir -p $OUT
export TMPDIR=/tmp
export GNNAGG_BUILD_LABEL=${GNNAGG_BUILD_LABEL:-$(md5sum gnn_computing_amd/libgnnagg.so | cut -c1-12)}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o t -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/trace_bench.log
cp $(find $OUT/trace_bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python3 scripts/bench_arms_from_trace.py $(find $OUT/trace_bench -name "*kernel_trace.csv" | head -1) $STEPS $WARM > $OUT/bench_arms.txt
for cfg in A ${CONFIGS:-R G P1}; do
  SQ=${SQ:-0} scripts/prof_config.sh $OUT $cfg > /dev/null 2>&1
done
python3 scripts/collect_profiles.py $OUT $OUT/pmc_traffic.json
ROUND=${ROUND:-r03}; mkdir -p profiles/$ROUND && cp $OUT/pmc_traffic.json profiles/$ROUND/pmc_traffic.json
python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
for cfg in ${CONFIGS:-R G P1}; do case $cfg in A_rows|P1_reorder) continue;; esac; python3 bench.py --config $cfg --no-cpu >> $OUT/bench_configs.jsonl 2>> $OUT/bench.err; done
rm -rf $OUT/trace_* $OUT/pmc_[A-Z]*_*
ls $OUT
