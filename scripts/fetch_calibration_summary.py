#!/usr/bin/env python3
"""Known bytes / counter for every case of scripts/micro/fetch_calibration.hip (scripts/fetch_calibration.sh OUTDIR).
Prints, per access pattern, the raw FETCH_SIZE / WRITE_SIZE (KB -> bytes), what the launch is known to have moved (useful bytes, in
64-B sectors, in 128-B lines) and the factor raw counter -> bytes for each of the three models; writes OUTDIR/fetch_calibration.json
(read by scripts/prof_config_summary.py and bench.py in place of the blanket x 2)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
known = {}
for line in open(os.path.join(out, "cal_run.txt")):
    if line.startswith("CAL "):
        f = line.split()
        known[f[1]] = {"useful": float(f[3]), "sector64": float(f[5]), "line128": float(f[7]), "ms": float(f[9])}
TAGS = {"1": "cal_g512_p512", "2": "cal_g256_p256", "3": "cal_g128_p128", "4": "cal_g400_p400", "5": "cal_g512_mall"}


def case_of(kernel):
    m = re.match(r".*cal_gather<\s*\d+\s*,\s*\d+\s*,\s*(\d+)\s*>", kernel)
    if m:
        return TAGS.get(m.group(1))
    for n in ("cal_stream16", "cal_write16", "cal_write512_rows"):
        if n in kernel:
            return n
    return None


pmc = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "cal_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        c = case_of(r["Kernel_Name"])
        if c:
            pmc[c][r["Counter_Name"]] = pmc[c].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = {}
for f in glob.glob(os.path.join(out, "cal_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        c = case_of(r["Kernel_Name"])
        if c:
            dur[c] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
res = {}
print("# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (one launch per case; gathers drawn uniformly from a 5 GB window unless named mall)")
print("# factor = known bytes / (counter x 1024); the guide's streaming case must come out at 2.0 for FETCH_SIZE")
for c in ("cal_stream16", "cal_g512_p512", "cal_g256_p256", "cal_g128_p128", "cal_g400_p400", "cal_g512_mall", "cal_write16", "cal_write512_rows"):
    k, p = known.get(c), pmc.get(c, {})
    if not k:
        continue
    wr = c.startswith("cal_write")
    raw = p.get("WRITE_SIZE" if wr else "FETCH_SIZE", 0.0) * 1024
    hit = p.get("TCC_HIT_sum", 0.0) / max(p.get("TCC_HIT_sum", 0.0) + p.get("TCC_MISS_sum", 0.0), 1.0)
    fac = {m: (k[m] / raw if raw else None) for m in ("useful", "sector64", "line128")}
    res[c] = {"known": k, "raw_counter_bytes": raw, "factor": fac, "l2_hit": hit, "kernel_us": dur.get(c), "counters": p}
    print("%-18s %s raw %8.3f GB | known useful %8.3f GB, 64-B sectors %8.3f GB, 128-B lines %8.3f GB | factor useful %.3f sector64 %.3f line128 %.3f | "
          "L2 hit %.3f | %.1f us" % (c, "WRITE_SIZE" if wr else "FETCH_SIZE", raw / 1e9, k["useful"] / 1e9, k["sector64"] / 1e9, k["line128"] / 1e9,
                                     fac["useful"] or 0, fac["sector64"] or 0, fac["line128"] or 0, hit, dur.get(c) or k["ms"] * 1e3))
    extra = {n: v for n, v in p.items() if n not in ("FETCH_SIZE", "WRITE_SIZE")}
    if extra:
        print("      " + "  ".join("%s=%.6g" % (n, v) for n, v in sorted(extra.items())))
json.dump(res, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
