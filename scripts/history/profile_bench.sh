#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun).
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, one group per run (no trace domains mixed in).
set -u
OUT=${1:-gpurun_out/prof}
CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu"
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUSY_avr"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -o pmc -- $CMD > $OUT/pmc_$tag.log 2>&1 || echo "pmc group failed: $grp"
done
ls -R $OUT | head -50
