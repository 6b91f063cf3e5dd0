#!/bin/bash
# per-kernel durations of the rows mode on the headline input, per library build and with / without the second stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in ${LIBS:-main}; do
  for aux in 1 0; do
    if [ "$lib" = main ]; then unset GNNAGG_LIB; else export GNNAGG_LIB=$R/gnn_computing_amd/csrc/build/ab/libgnnagg_$lib.so; fi
    export ROWS_AUX=$aux ROWS_MEDIUM_SET=256
    d=$R/gpurun_out/rowsprobe/${lib}_aux$aux
    rm -rf $d; mkdir -p $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/scripts/exp_rows_mode.py A > $d/out.txt 2>/dev/null
    echo "== $lib aux=$aux  $(python3 -c "import json;d=json.loads(open('$d/out.txt').read().strip().splitlines()[-1]);print('rows_us %.1f' % d['rows_us'])")"
    python3 - <<PY
import csv,glob
f=glob.glob('$d/**/t_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'gnnagg' in r['Name']: print('   %-60s n=%s avg %.1f us min %.1f max %.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  done
done
