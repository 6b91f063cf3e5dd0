#!/usr/bin/env python3
"""Arxiv-shaped input across feature widths and run modes (one line per case): for A/B builds through GNNAGG_LIB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def t(fn, warm=5, iters=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


ptr, idx = gnc.graph.dataset("arxiv", device=dev)
if os.environ.get("REORDER") == "1":   # the locality reorder bench.py applies on load
    p, i = ptr.cpu().numpy(), idx.cpu().numpy()
    rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    p, i, _ = gnc.reorder_csr(p, i, rows)
    ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
V, E = ptr.numel() - 1, idx.numel()
val = torch.ones(E, device=dev)
out = []
for F in (16, 32, 64, 100, 128, 256, 512):
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
    agg.schedule(gnc.Schedule.neighbor_grouping, [32])
    out.append("F=%-3d rows %7.1f  ng32 %7.1f  balanced %7.1f  mean %7.1f  max %7.1f" % (
        F, t(lambda: agg.run(x, y, 512, 0)), t(lambda: agg.run(x, y, 512, 1)), t(lambda: agg.run(x, y, 512, "balanced")),
        t(lambda: agg.run(x, y, 512, "balanced", reduce="mean")), t(lambda: agg.run(x, y, 512, "balanced", reduce="max"))))
for F, H in ((32, 1), (128, 1), (256, 8)):
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    att = torch.randn((V, H, 2), device=dev) * 0.3
    gat = gnc.Aggregator_GAT(ptr, idx, F, F)
    gat.schedule(gnc.Schedule.neighbor_grouping, [32])
    out.append("GAT F=%-3d H=%d rows %7.1f  ng32 %7.1f  balanced %7.1f" % (
        F, H, t(lambda: gat.run(x, att, y, 128, 0, heads=H)), t(lambda: gat.run(x, att, y, 128, 1, heads=H)),
        t(lambda: gat.run(x, att, y, 128, "balanced", heads=H))))
print("\n".join(out))
