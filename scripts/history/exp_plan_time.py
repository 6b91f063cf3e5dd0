#!/usr/bin/env python3
"""Construction time of the library-chosen blocked orders (the reference prints its own schedule time,
graph_schedule.h:125-127): reddit-shaped graph, balanced order and the rows mode's chain plan, device builder (plan_gpu.hip) against
the host builders of rounds 2-3 ("host_plan" = 1); first run and steady-state step beside them; results compared bit for bit.
usage: exp_plan_time.py [reddit|products]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset(name, device=dev)
V, E = ptr.numel() - 1, idx.numel()
F = 602 if name == "reddit" else 100
x = torch.randn((V, F), device=dev)
outs = {}
for host in (0, 1):
    for mode, kw in (("balanced", {"reduce": "mean"}), (0, {"reduce": "mean"})):
        agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
        if name != "reddit":
            agg.set_option("partitions", 16)   # (products-shaped: the library would not block it; forced, to time the builder on 2.4 M rows)
        agg.set_option("host_plan", host)
        y = torch.empty((V, F), device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agg.run(x, y, 512, mode, **kw)     # first call: plan + scratch + the step
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(5):
            agg.run(x, y, 512, mode, **kw)
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / 5
        info = agg.plan_info()
        key = "balanced" if mode == "balanced" else "rows"
        outs.setdefault(key, []).append(y.clone())
        print("%s %-8s builder=%s: plan %.3f s (rows-mode chain plan %.3f s), first call %.3f s, step %.2f ms, plan arrays %.2f GB, scratch %.2f GB, ranges %d" % (
            name, key, "host" if host else "device", info["plan_s"], info["rows_plan_s"], t_first, t_step * 1e3, info["plan_bytes"] / 1e9,
            info["scratch_bytes"] / 1e9, agg.balanced_partitions() if mode == "balanced" else agg.rows_blocked_ranges()), flush=True)
        del agg, y
        torch.cuda.empty_cache()
for k, (a, b) in outs.items():
    print("%s: device-built and host-built plans give %s results" % (k, "bit-equal" if torch.equal(a, b) else "DIFFERENT"))
