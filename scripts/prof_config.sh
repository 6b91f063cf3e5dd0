#!/bin/bash
# usage: scripts/prof_config.sh OUTDIR CFG [ENV=VAL ...]  -- rocprofv3 kernel trace + PMC passes (traffic, L2 hit rate) of a few
# balanced-mode launches of one named config (scripts/run_config_once.py R|G|P1); per-kernel summary -> OUTDIR/summary_CFG.txt
set -u
OUT=$1; CFG=$2; shift 2
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -o t -- python3 scripts/run_config_once.py $CFG > $OUT/trace_$CFG.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${CFG}_$tag -o pmc -- python3 scripts/run_config_once.py $CFG > $OUT/pmc_${CFG}_$tag.log 2>&1 || echo "pmc group failed: $grp"
done
python3 scripts/prof_config_summary.py $OUT $CFG "$*" | tee $OUT/summary_$CFG.txt
