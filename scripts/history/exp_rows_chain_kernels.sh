#!/bin/bash
# per-kernel durations of the chained rows mode (R, G) beside the balanced mode: where the gap between the two goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${CFGS:-R G}; do
  d=$R/gpurun_out/rowschain/$cfg
  rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/scripts/exp_rows_mode.py $cfg > $d/out.txt 2>/dev/null
  echo "== $cfg $(tail -1 $d/out.txt | cut -c1-300)"
  python3 - <<PY
import csv,glob
f=glob.glob('$d/**/t_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'gnnagg' in r['Name']: print('   %-70s n=%5s avg %9.1f us  total %9.1f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
