#!/bin/bash
# usage: scripts/prof_config.sh OUTDIR CFG [ENV=VAL ...]  -- rocprofv3 kernel trace + PMC passes (traffic, L2 hit rate) of a few
# balanced-mode launches of one named config (scripts/run_config_once.py R|G|P1); per-kernel summary -> OUTDIR/summary_CFG.txt
set -u
OUT=$1; CFG=$2; shift 2
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -o t -- python3 scripts/run_config_once.py $CFG > $OUT/trace_$CFG.log 2>&1
PMCG=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum")
if [ "${SQ:-0}" = "1" ]; then
  PMCG+=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE")
fi
for grp in "${PMCG[@]}"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${CFG}_$tag -o pmc -- python3 scripts/run_config_once.py $CFG > $OUT/pmc_${CFG}_$tag.log 2>&1 || echo "pmc group failed: $grp"
done
python3 scripts/prof_config_summary.py $OUT $CFG "$*" | tee $OUT/summary_$CFG.txt
