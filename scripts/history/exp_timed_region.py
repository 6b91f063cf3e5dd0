#!/usr/bin/env python3
"""Where do the ~6 us per step go that bench.py's wall clock shows at --steps 20 beyond the 74 us launches?  Variants of the timed
region around the same 20 launches of the headline aggregation (arxiv-shaped, F = 128, balanced order, reorder applied)."""
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda:0")
ptr, idx = gnc.graph.dataset("arxiv")
rows, _ = gnc.cluster_reorder(ptr.numpy(), idx.numpy(), order="cache_greedy", cluster_cap=1, cache_rows=8192)
p2, i2, _ = gnc.reorder_csr(ptr.numpy(), idx.numpy(), np.asarray(rows, np.int32))
V, E = len(p2) - 1, len(i2)
agg = gnc.Aggregator_GCN(torch.from_numpy(p2).to(dev), torch.from_numpy(i2).to(dev), torch.ones(E, device=dev), 128, 128)
x = torch.randn(V, 128, device=dev)
y = torch.empty_like(x)
step = lambda: agg.run(x, y, 512, "balanced")
K, W = 20, 5


def region(sleep_s, warm_after_sleep, events, reps=7):
    out = []
    for _ in range(reps):
        for _ in range(W):
            step()
        torch.cuda.synchronize()
        gc.collect()
        if sleep_s:
            time.sleep(sleep_s)
        if warm_after_sleep:
            for _ in range(W):
                step()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if events:
            ev0.record()
        for _ in range(K):
            step()
        if events:
            ev1.record()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e6)
    out.sort()
    return out[len(out) // 2], out[0], out[-1]


for name, kw in (("as bench.py does it (sleep 0.15 s after the warm-up, events)", dict(sleep_s=0.15, warm_after_sleep=False, events=True)),
                 ("warm-up after the sleep", dict(sleep_s=0.15, warm_after_sleep=True, events=True)),
                 ("no sleep", dict(sleep_s=0.0, warm_after_sleep=False, events=True)),
                 ("no sleep, no events", dict(sleep_s=0.0, warm_after_sleep=False, events=False)),
                 ("warm-up after the sleep, no events", dict(sleep_s=0.15, warm_after_sleep=True, events=False))):
    med, lo, hi = region(**kw)
    print("%-62s wall per step: median %.2f us (min %.2f, max %.2f)" % (name, med, lo, hi), flush=True)
