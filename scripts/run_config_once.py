#!/usr/bin/env python3
"""A few balanced-mode launches of one named config (R, G or P1) -- the command rocprofv3 wraps for per-config counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "R"
dev = torch.device("cuda", 0)
name, F = {"A": ("arxiv", 128), "A_rows": ("arxiv", 128), "R": ("reddit", 602), "G": ("reddit", 256), "P1": ("products", 100)}[cfg]
mode = "rows" if cfg == "A_rows" else "balanced"   # A_rows: the headline input in GNNAGG_MODE_ROWS (the literal aggr_gcn order)
if cfg in ("A", "A_rows"):   # the headline workload of bench.py: arxiv-shaped, locality reorder applied on load, explicit unit weights
    import numpy as np
    p, i = gnc.graph.dataset(name)
    p, i = p.numpy(), i.numpy()
    rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    p, i, _ = gnc.reorder_csr(p, i, rows)
    ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
else:
    ptr, idx = gnc.graph.dataset(name, device=dev)
V = ptr.numel() - 1
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)
if cfg == "G":
    agg = gnc.Aggregator_GAT(ptr, idx, F, F)
    att = torch.randn((V, 8, 2), device=dev) * 0.3
    agg.schedule_balanced(0)
    run = lambda: agg.run(x, att, y, 128, "balanced", heads=8)  # noqa: E731
else:
    agg = gnc.Aggregator_GCN(ptr, idx, torch.ones(idx.numel(), device=dev) if cfg in ("A", "A_rows", "P1") else None, F, F)
    if mode == "balanced":
        agg.schedule_balanced(0)
    run = lambda: agg.run(x, y, 128, mode, reduce="mean" if cfg == "R" else "sum")  # noqa: E731
for _ in range(20 if cfg in ("A", "A_rows") else 5):
    run()
torch.cuda.synchronize()
