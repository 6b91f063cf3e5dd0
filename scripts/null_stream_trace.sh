#!/bin/bash
# usage (GPU box): scripts/null_stream_trace.sh OUTDIR -> OUTDIR/null_stream_trace.txt: per phase, average kernel duration and start-to-start spacing
set -u
OUT=${1:-gpurun_out/null_trace}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 scripts/null_stream_trace.py > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
plan = [k for k in ks if "k_gcn_plan" in k[0]]
# the phases are separated by the largest gap between consecutive k_gcn_plan launches
gaps = [plan[i + 1][1] - plan[i][2] for i in range(len(plan) - 1)]
cut = gaps.index(max(gaps)) + 1
lines = []
for name, ph in (("null stream, the process has one queue", plan[:cut]), ("null stream, after a second stream ran a kernel", plan[cut:])):
    ph = ph[-100:]
    dur = sum(e - s for _, s, e in ph) / len(ph) / 1e3
    spacing = (ph[-1][1] - ph[0][1]) / (len(ph) - 1) / 1e3
    gap = sum(ph[i + 1][1] - ph[i][2] for i in range(len(ph) - 1)) / (len(ph) - 1) / 1e3
    lines.append("%-50s n=%d  kernel duration %.2f us  start-to-start %.2f us  end-to-next-start %+.2f us" % (name, len(ph), dur, spacing, gap))
open(out + "/null_stream_trace.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
