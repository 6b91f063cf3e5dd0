#!/usr/bin/env python3
"""One arm of the locality experiment for rocprofv3: exp_locality_one.py <plain|community> <par|0> [F]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
comm = sys.argv[1] == "community"
par = int(sys.argv[2])
F = int(sys.argv[3]) if len(sys.argv) > 3 else 602
V, E = gnc.graph.SHAPES[os.environ.get("EXP_DATASET", "reddit")]
ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev, community_order=comm)
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
if par > 0:
    agg.schedule(gnc.Schedule.locality, [par])
    run = lambda: agg.run(x, y, 128, 1)  # noqa: E731
else:
    agg.schedule_balanced(0)
    run = lambda: agg.run(x, y, 128, "balanced")  # noqa: E731
for _ in range(4):
    run()
torch.cuda.synchronize()
