#!/usr/bin/env python3
"""Runs one aggregation mode repeatedly on the arxiv-shaped input (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rows"
kind = sys.argv[2] if len(sys.argv) > 2 else "gcn"
F = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("arxiv")
ptr, idx = ptr.to(dev), idx.to(dev)
V = ptr.numel() - 1
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
if kind == "gcn":
    agg = gnc.Aggregator_GCN(ptr, idx, torch.ones(idx.numel(), device=dev), F, F)
    run = lambda: agg.run(x, y, 512, mode)  # noqa: E731
else:
    agg = gnc.Aggregator_GAT(ptr, idx, F, F)
    att = torch.randn((V, 2), device=dev)
    run = lambda: agg.run(x, att, y, 128, mode)  # noqa: E731
if mode == "scheduled":
    agg.schedule(gnc.Schedule.neighbor_grouping, [32])
for _ in range(30):
    run()
torch.cuda.synchronize()
