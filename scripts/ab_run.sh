#!/bin/bash
# scripts/ab_run.sh "<lib names: main or build/ab names>" "<CONFIGS>" : runs scripts/bench_configs.py once per library build.
cd "$(dirname "$0")/.."
for n in $1; do
  if [ "$n" = main ]; then unset GNNAGG_LIB; else export GNNAGG_LIB=$PWD/gnn_computing_amd/csrc/build/ab/libgnnagg_$n.so; fi
  echo "== $n"
  CONFIGS=$2 MODES=balanced ITERS=${ITERS:-10} python3 scripts/bench_configs.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-32s %9.1f us  %.2f G edges/s' % (d['config'], d['us'], d['edges_per_s'] / 1e9))"
done
