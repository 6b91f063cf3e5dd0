#!/usr/bin/env python3
"""Per-kernel, per-counter dispatch averages of the rocprofv3 --pmc passes written by scripts/profile_bench.sh, and the
profiles/pmc_traffic.json that bench.py reports as roofline.traffic.
usage: pmc_summary.py <prof dir> <out txt> [<out traffic json>]"""
import collections
import csv
import glob
import json
import os
import sys

prof, out_txt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(prof, "pmc_*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "gnnagg" not in r["Kernel_Name"]:
            continue
        a = acc[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
lines = ["# rocprofv3 --pmc passes over `python3 bench.py --steps 50 --warmup 5 --no-cpu` (scripts/profile_bench.sh), per-dispatch averages",
         "# dispatches = (50 timed + 5 warm-up) x (no-reorder arm + reorder_thres_0.2 arm)"]
avg = {}
for (k, c), (s, n) in sorted(acc.items(), key=lambda t: (t[0][1], t[0][0])):
    avg[(k, c)] = s / n
    lines.append("%-48s %-30s n=%4d avg=%16.1f" % (k[:48], c, n, s / n))
open(out_txt, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if len(sys.argv) > 3:
    plan = [k for (k, c) in avg if "k_gcn_plan<" in k and c == "FETCH_SIZE"][0]
    fetch, write = avg[(plan, "FETCH_SIZE")], avg[(plan, "WRITE_SIZE")]
    rd = avg.get((plan, "TCC_EA0_RDREQ_sum"))
    json.dump({
        "_source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 50 "
                   "--warmup 5 --no-cpu` (scripts/profile_bench.sh, scripts/pmc_summary.py), per-dispatch average over both "
                   "arms (no-reorder + reorder_thres_0.2)" % out_txt,
        "_correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE (KB) counts the 128-B requests of a wide "
                       "coalesced 16 B/lane read at 64 B -> doubled; cross-checked with TCC_EA0_RDREQ_sum = %s requests x "
                       "128 B. WRITE_SIZE taken as reported (KB x 1024). These fabric-side counters include Infinity-Cache "
                       "hits." % ("%.4g" % rd if rd else "n/a"),
        "kernel": plan, "fetch_size_kb": fetch, "write_size_kb": write,
        "hbm_bytes_per_launch": int(round(fetch * 1024 * 2 + write * 1024)),
    }, open(sys.argv[3], "w"), indent=1)
