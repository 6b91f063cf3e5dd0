#!/usr/bin/env python3
"""Stress of the in-kernel hub fold (arrival counters + device-scope scratch traffic across XCDs): thousands of launches
alternating inputs on two handles and two streams, every result compared bit-for-bit with the first launch's."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
V, E, F = 20000, 1500000, 128
ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=3, alpha=1.1, device=dev)
val = torch.randn(E, device=dev)
aggs = [gnc.Aggregator_GCN(ptr, idx, val, F, F) for _ in range(2)]
for a in aggs:
    a.schedule_balanced(16)     # 256-edge segments: dozens of hubs with tens to hundreds of segments
xs = [torch.randn((V, F), device=dev) for _ in range(3)]
refs = []
y = torch.empty((V, F), device=dev)
for x in xs:
    aggs[0].run(x, y, 128, "balanced")
    refs.append(y.clone())
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ys = [torch.empty((V, F), device=dev) for _ in range(2)]
bad = 0
N = int(os.environ.get("STRESS_ITERS", "3000"))
for it in range(N):
    k = it % 3
    for a, st, yy in zip(aggs, streams, ys):
        with torch.cuda.stream(st):
            a.run(xs[k], yy, 128, "balanced")
    if it % 50 == 0 or it > N - 5:
        torch.cuda.synchronize()
        for yy in ys:
            if not torch.equal(yy, refs[k]):
                bad += 1
torch.cuda.synchronize()
deg = (ptr[1:] - ptr[:-1])
print("launches: %d x 2 handles, hubs (> 4096 edges): %d, max degree %d, mismatching checks: %d" % (
    N, int((deg > 4096).sum()), int(deg.max()), bad))
sys.exit(1 if bad else 0)
