#!/usr/bin/env python3
"""A few balanced-mode launches of one named config (A, A_rows, R, G, P1, P1_reorder) -- the command rocprofv3 wraps for per-config counters.
Launched on ONE NON-NULL stream, like bench.py (null-stream launches slow down once a process has other streams: DESIGN.md section 5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "R"
dev = torch.device("cuda", 0)
name, F = {"A": ("arxiv", 128), "A_rows": ("arxiv", 128), "R": ("reddit", 602), "G": ("reddit", 256), "P1": ("products", 100),
           "P1_reorder": ("products", 100)}[cfg]
mode = "rows" if cfg == "A_rows" else "balanced"   # A_rows: the headline input in GNNAGG_MODE_ROWS (the literal aggr_gcn order)
if cfg in ("A", "A_rows", "P1_reorder"):   # bench.py's arms with the locality reorder applied on load (same cache file: same library key)
    import hashlib
    key = hashlib.md5(open(os.environ.get("GNNAGG_LIB") or gnc._lib.LIB_PATH, "rb").read()).hexdigest()[:12]
    p, i = gnc.graph.dataset(name, device=dev if cfg == "P1_reorder" else "cpu")
    p, i, _, _, _ = gnc.graph.reorder_on_load(name, p.cpu().numpy(), i.cpu().numpy(), key=key)
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
    agg = gnc.Aggregator_GCN(ptr, idx, torch.ones(idx.numel(), device=dev) if cfg in ("A", "A_rows", "P1", "P1_reorder") else None, F, F)
    if mode == "balanced":
        agg.schedule_balanced(0)
    run = lambda: agg.run(x, y, 128, mode, reduce="mean" if cfg == "R" else "sum")  # noqa: E731
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    for _ in range(20 if cfg in ("A", "A_rows") else 5):
        run()
torch.cuda.synchronize()
