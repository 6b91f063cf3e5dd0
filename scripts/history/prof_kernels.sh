#!/bin/bash
# usage: scripts/prof_kernels.sh TAG <python args...>   -- rocprofv3 kernel stats of one python command, top kernels printed
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o t -- python3 "$@" > gpurun_out/$TAG.log 2>&1
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
echo "== $TAG"; head -7 $f | cut -d, -f1-4 | cut -c1-120
