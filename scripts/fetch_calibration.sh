#!/bin/bash
# usage: scripts/fetch_calibration.sh OUTDIR  -- known-byte-count launches of scripts/micro/fetch_calibration.hip under rocprofv3 --pmc
# (separate passes, the program directly after --), then known bytes / counter per access pattern -> OUTDIR/fetch_calibration.txt
set -u
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
EXE=scripts/micro/fetch_calibration.out
[ -x $EXE ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calibration.hip -o $EXE
$EXE > $OUT/cal_run.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*MALL[A-Z_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE[A-Z_]*" | sort -u > $OUT/cal_counters_available.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cal_trace -o t -- $EXE > $OUT/cal_trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_READ_sum TCC_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/cal_pmc_$tag -o pmc -- $EXE > $OUT/cal_pmc_$tag.log 2>&1 || echo "pmc group failed: $grp"
done
python3 scripts/fetch_calibration_summary.py $OUT | tee $OUT/fetch_calibration.txt
