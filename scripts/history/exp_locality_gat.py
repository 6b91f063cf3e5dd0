#!/usr/bin/env python3
"""Experiment: locality schedules for the GAT config (reddit-shaped, 8 heads x 32) and products-shaped GCN F=100."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


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


V, E = gnc.graph.SHAPES["reddit"]
ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev)
F, H = 256, 8
x, att = torch.randn((V, F), device=dev), torch.randn((V, H, 2), device=dev) * 0.3
y = torch.empty((V, F), device=dev)
gat = gnc.Aggregator_GAT(ptr, idx, F, F)
gat.schedule_balanced(0)
print("G balanced: %.2f ms" % t(lambda: gat.run(x, att, y, 128, "balanced", heads=H)), flush=True)
yb = y.clone()
for par, ng in ((8, 0), (16, 0), (16, 512), (24, 0)):
    if ng:
        gat.schedule(gnc.Schedule.locality_neighbor_grouping, [par, ng])
    else:
        gat.schedule(gnc.Schedule.locality, [par])
    ms = t(lambda: gat.run(x, att, y, 128, 1, heads=H))
    print("   G locality par=%d ng=%d: %.2f ms (max rel diff %.1e)" % (par, ng, ms, float((y - yb).abs().max() / yb.abs().max())), flush=True)
del gat, x, att, y, yb, ptr, idx
V, E = gnc.graph.SHAPES["products"]
ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev)
F = 100
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
agg.schedule_balanced(0)
print("P1 balanced: %.2f ms" % t(lambda: agg.run(x, y, 128, "balanced")), flush=True)
for par in (8, 16):
    agg.schedule(gnc.Schedule.locality, [par])
    print("   P1 locality par=%d: %.2f ms" % (par, t(lambda: agg.run(x, y, 128, 1))), flush=True)
