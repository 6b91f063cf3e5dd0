#!/usr/bin/env python3
"""Config G in GNNAGG_MODE_ROWS on the blocked order: hub threshold x slice size sweep (options "rows_hub_edges", "slice_kb", "partitions")."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("reddit", device=dev)
V, E = ptr.numel() - 1, idx.numel()
F, H = 256, 8
x, att = torch.randn((V, F), device=dev), torch.randn((V, H, 2), device=dev) * 0.4
kind = sys.argv[1] if len(sys.argv) > 1 else "gat"
PARTS = [int(v) for v in os.environ.get("SWEEP_PARTS", "8,15,23").split(",")]
HUBS = [int(v) for v in os.environ.get("SWEEP_HUBS", "512,1024,2048,4096,16384").split(",")]
for parts in PARTS:
    for hub in HUBS:
        g = gnc.Aggregator_GAT(ptr, idx, F, F) if kind == "gat" else gnc.Aggregator_GCN(ptr, idx, None, F, F)
        g.set_option("partitions", parts)
        g.set_option("rows_hub_edges", hub)
        y = torch.empty((V, F), device=dev)
        run = (lambda: g.run(x, att, y, 128, 0, heads=H)) if kind == "gat" else (lambda: g.run(x, y, 128, 0))
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        print("%s rows mode, %2d ranges, hub rows above %5d edges per (row, range): %7.2f ms" % (kind, g.rows_blocked_ranges(), hub, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
        del g
