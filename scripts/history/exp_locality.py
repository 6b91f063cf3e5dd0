#!/usr/bin/env python3
"""Experiment: source-partitioned (locality) schedules vs the balanced mode on the high-degree configs, plain and community order."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
name = os.environ.get("EXP_DATASET", "reddit")
F = int(os.environ.get("EXP_F", "602"))
V, E = gnc.graph.SHAPES[name]


def t(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for comm in (False, True):
    ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev, community_order=comm)
    x = torch.randn((V, F), device=dev)
    y = torch.empty((V, F), device=dev)
    agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    agg.schedule_balanced(0)
    ms = t(lambda: agg.run(x, y, 128, "balanced"))
    print("%s %s F=%d balanced: %.2f ms" % (name, "community" if comm else "plain", F, ms), flush=True)
    yb = y.clone()
    for par in [int(v) for v in os.environ.get("EXP_PARS", "8,16,32").split(",")]:
        t0 = time.time()
        agg.schedule(gnc.Schedule.locality, [par])
        prep = time.time() - t0
        ms = t(lambda: agg.run(x, y, 128, 1))
        err = float((y - yb).abs().max() / yb.abs().max())
        print("   locality par=%d: %.2f ms (schedule %.1f s, max rel diff vs balanced %.1e)" % (par, ms, prep, err), flush=True)
