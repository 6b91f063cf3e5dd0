#!/usr/bin/env python3
"""Experiment: how much of a locality order's gain is the LAYOUT (a row's sources sit next to each other in memory) and how
much the TIME order (rows processed together share sources)?  The generator's community order, with the node ids shuffled
at random inside blocks of B consecutive positions: coarse locality (which rows run together, which X rows they touch) is
kept, the fine layout is destroyed."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
V, E = gnc.graph.SHAPES["arxiv"]
F = 128
p, i = gnc.graph.powerlaw_csr(V, E, seed=123, community_order=True)
ptr, idx = p.numpy(), i.numpy()
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)


def run(ptr, idx, tag):
    agg = gnc.Aggregator_GCN(torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.ones(len(idx), device=dev), F, F)
    for _ in range(20):
        agg.run(x, y, 512, "balanced")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        agg.run(x, y, 512, "balanced")
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({"order": tag, "us": a.elapsed_time(b) * 1e3 / 200}), flush=True)


run(ptr, idx, "community")
rng = np.random.default_rng(0)
for B in (64, 512, 4096, 32768):
    rows = np.arange(V, dtype=np.int32)
    for b0 in range(0, V, B):
        rng.shuffle(rows[b0:b0 + B])
    np_, ni_, _ = gnc.reorder_csr(ptr, idx, rows)
    run(np_, ni_, "community, ids shuffled inside blocks of %d" % B)
rows = rng.permutation(V).astype(np.int32)
np_, ni_, _ = gnc.reorder_csr(ptr, idx, rows)
run(np_, ni_, "fully shuffled")
