#!/bin/bash
# scripts/ab_bench.sh "<lib names: main or build/ab names>" : the headline bench line (bench.py, no CPU leg) once per library build
cd "$(dirname "$0")/.."
for n in $1; do
  if [ "$n" = main ]; then unset GNNAGG_LIB; else export GNNAGG_LIB=$PWD/gnn_computing_amd/csrc/build/ab/libgnnagg_$n.so; fi
  python3 bench.py --steps 200 --warmup 20 --no-cpu 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('  %-10s headline %.2f us  probe %.2f us  frac %.3f  no_reorder %.2f us' % ('$n', d['ms_per_step'] * 1e3, r['ceiling_probe_us'], r['frac'], d['no_reorder']['avg_launch_us']))"
done
