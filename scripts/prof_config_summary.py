#!/usr/bin/env python3
"""Per-kernel summary of scripts/prof_config.sh: average duration (kernel trace), fabric-side traffic (FETCH_SIZE x 2 per the
gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, KB -> bytes) and L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS)."""
import collections
import csv
import glob
import os
import sys

out, cfg = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "trace_" + cfg, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[r["Kernel_Name"]]
        d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        d[1] += 1
pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc_%s_*" % cfg, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = pmc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
import json
import subprocess
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    commit = os.environ.get("GNNAGG_BUILD_LABEL", "unknown")
kernels = []
PLAN_KERNELS = ("k_range_keys", "k_mark_row_starts", "k_rows_unsorted", "k_gather_sorted", "k_subrow_starts", "k_group_flags", "k_scatter_groups",
                "k_count_groups", "k_iota", "k_flag_ids", "k_mark_hub_rows", "k_chain_", "k_gather_u32",
                "prims::")   # prims:: = the plan builder's scan / sort / reduce primitives (prims.cuh): once per handle too
print("# config %s %s -- per-launch averages; traffic = FETCH_SIZE*2*1024 + WRITE_SIZE*1024 (fabric side, Infinity-Cache hits included)" % (cfg, sys.argv[3] if len(sys.argv) > 3 else ""))
tot = 0.0
for k, (s, n) in sorted(dur.items(), key=lambda t: -t[1][0]):
    if "gnnagg" not in k or any(n in k for n in PLAN_KERNELS):   # (the plan builder's kernels run once per handle, not once per step)
        continue
    avg = s / n
    c = {name: v[0] / v[1] for name, v in pmc.get(k, {}).items()}
    traffic = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024
    hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    print("%-70s n=%3d avg %10.1f us  traffic %8.2f GB (%6.2f TB/s)  fetch %8.2f GB write %8.2f GB  L2 hit %.3f  tcp->tcc rd %.3g  ea rd %.3g" % (
        k[:70], n, avg, traffic / 1e9, traffic / avg / 1e6 if avg else 0, c.get("FETCH_SIZE", 0) * 2048 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9, hit,
        c.get("TCP_TCC_READ_REQ_sum", 0), c.get("TCC_EA0_RDREQ_sum", 0)))
    kernels.append({"kernel": k, "launches": n, "avg_us": avg, "traffic_bytes": traffic, "fetch_bytes": c.get("FETCH_SIZE", 0) * 2048,
                    "write_bytes": c.get("WRITE_SIZE", 0) * 1024, "l2_hit": hit})
    if c.get("TCC_EA0_RDREQ_DRAM_sum") is not None and c.get("TCC_EA0_RDREQ_sum"):
        # read requests that went on to DRAM (the rest of the fabric-side requests were served by the Infinity Cache); 128-B requests
        print("      ea rd to DRAM %.4g of %.4g requests (%.3f) -> HBM fetch ~ %.2f GB of the %.2f GB fabric fetch; 128-B requests %.4g" % (
            c["TCC_EA0_RDREQ_DRAM_sum"], c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_DRAM_sum"] / c["TCC_EA0_RDREQ_sum"],
            c.get("FETCH_SIZE", 0) * 2048 / 1e9 * c["TCC_EA0_RDREQ_DRAM_sum"] / c["TCC_EA0_RDREQ_sum"], c.get("FETCH_SIZE", 0) * 2048 / 1e9,
            c.get("TCC_EA0_RDREQ_128B_sum", 0)))
    extra = {n: v for n, v in c.items() if n.startswith(("SQ_", "TA_", "GRBM", "TCP_TOTAL"))}
    if extra:
        print("      " + "  ".join("%s=%.4g" % (n, v) for n, v in sorted(extra.items())))
    tot += avg
print("sum of per-launch averages (the kernels of ONE step; plan-time kernels excluded): %.1f us" % tot)
json.dump({"config": cfg, "options": sys.argv[3] if len(sys.argv) > 3 else "", "build": os.environ.get("GNNAGG_BUILD_LABEL", commit),
           "_correction": "FETCH_SIZE (KB) x 2 x 1024 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE "
                          "(KB) x 1024; fabric-side counters, Infinity-Cache hits included; separate --pmc passes",
           "kernels": kernels}, open(os.path.join(out, "summary_%s.json" % cfg), "w"), indent=1)
