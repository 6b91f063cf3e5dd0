#!/usr/bin/env python3
"""profiles/<round>/pmc_traffic.json from the per-config summaries of scripts/prof_config.sh: for every config the
dominant kernel's fabric-side bytes per launch (what bench.py reports as roofline.traffic, labelled with the build) and the
per-launch totals.  usage: collect_profiles.py <dir with summary_*.json> <out json>"""
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(os.path.join(src, "summary_*.json"))):
    d = json.load(open(f))
    # (summary_*.json lists the kernels of a STEP only: prof_config_summary.py drops the plan builder's, which run once per handle)
    ks = [k for k in d["kernels"] if k["launches"] > 0 and "prims::" not in k["kernel"]]
    if not ks:
        continue
    dom = max(ks, key=lambda k: k["avg_us"])
    res[d["config"]] = {
        "_label": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/run_config_once.py %s, build %s; %s" % (
            d["config"], d["build"], d["_correction"]),
        "build": d["build"], "kernel": dom["kernel"], "avg_us": dom["avg_us"], "l2_hit": dom["l2_hit"],
        "hbm_bytes_per_launch": int(dom["traffic_bytes"]),
        "all_kernels_bytes_per_step": int(sum(k["traffic_bytes"] for k in ks)),
        "all_kernels_us_per_step": sum(k["avg_us"] for k in ks),
        "kernels": [{"kernel": k["kernel"][:90], "avg_us": k["avg_us"], "traffic_bytes": int(k["traffic_bytes"]), "l2_hit": k["l2_hit"]} for k in ks],
    }
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v["kernel"][:50], v["avg_us"], v["hbm_bytes_per_launch"]) for k, v in res.items()}))
