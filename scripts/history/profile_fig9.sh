#!/bin/bash
# Figure-9-style evidence: L2 hit rate and duration of the aggregation kernel with / without the locality reorder.
OUT=${1:-gpurun_out/fig9}
mkdir -p $OUT
export TMPDIR=/tmp
ARMS=${ARMS:-"none lsh greedy community"}
for arm in $ARMS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$arm -o t -- python3 scripts/run_reorder_arm.py $arm > $OUT/trace_$arm.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$arm -o p -- python3 scripts/run_reorder_arm.py $arm > $OUT/pmc_$arm.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$arm -o p -- python3 scripts/run_reorder_arm.py $arm > $OUT/fetch_$arm.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for arm in "$ARMS".split():
    st = [r for r in csv.DictReader(open("$OUT/trace_%s/t_kernel_stats.csv" % arm)) if "k_gcn_plan" in r["Name"]][0]
    agg = collections.defaultdict(list)
    for f in ("$OUT/pmc_%s/p_counter_collection.csv" % arm, "$OUT/fetch_%s/p_counter_collection.csv" % arm):
        for r in csv.DictReader(open(f)):
            if "k_gcn_plan" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print("%-16s k_gcn_plan avg %.1f us | L2 hit rate %.3f | FETCH_SIZE %.0f KB (x2 = %.0f MB from the fabric)" % (
        arm, float(st["AverageNs"]) / 1e3, m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), m["FETCH_SIZE"],
        2 * m["FETCH_SIZE"] * 1024 / 1e6))
PY
