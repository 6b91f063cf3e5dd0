#!/usr/bin/env python3
"""One arm of the with/without-reorder comparison (reference Figure9/run.sh measures l2_tex_hit_rate of the
aggregation kernel with and without `--reorder _thres_0.2`): runs the balanced GCN aggregation on the arxiv-shaped
input, either as generated ("none") or after the MinHash-LSH clustering reorder ("lsh") / in community order
("community").  Meant to run under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

arm = sys.argv[1] if len(sys.argv) > 1 else "none"
F = 128
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("arxiv")
ptr, idx = ptr.numpy(), idx.numpy()
if arm == "lsh":
    rows, _ = gnc.cluster_reorder(ptr, idx)
    ptr, idx, _ = gnc.reorder_csr(ptr, idx, rows)
elif arm.startswith("greedy"):   # greedy[:cache_rows[:cluster_cap]]
    parts = arm.split(":")
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cache_rows=int(parts[1]) if len(parts) > 1 else 4096,
                                  cluster_cap=int(parts[2]) if len(parts) > 2 else 1)
    ptr, idx, _ = gnc.reorder_csr(ptr, idx, rows)
elif arm == "community":
    V, E = gnc.graph.SHAPES["arxiv"]
    p, i = gnc.graph.powerlaw_csr(V, E, seed=123, community_order=True)
    ptr, idx = p.numpy(), i.numpy()
dptr, didx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
V = len(ptr) - 1
agg = gnc.Aggregator_GCN(dptr, didx, torch.ones(len(idx), device=dev), F, F)
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
for _ in range(40):
    agg.run(x, y, 512, "balanced")
torch.cuda.synchronize()
