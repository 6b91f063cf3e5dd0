#!/usr/bin/env python3
"""P1 (products-shaped GCN F=100, one GPU) with the locality reorder applied on load.  The cache-aware greedy order of this
124 M-edge graph (gnnagg_cluster_reorder_ex, order_mode 1, cluster_cap 1, 8192 cache rows) took 5.4 minutes of one core in
round 2; round 3 runs it with one walker per thread (reorder.cpp, emit_cache_greedy_parallel): computed here, on the box's host
cores, and timed.  GNNAGG_REORDER_WALKERS=1 gives the serial pass.  The graph is generated on the CPU (the CPU and GPU
generators of torch draw different graphs for the same seed)."""
import time
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
F = 100
p, i = gnc.graph.dataset("products")
ptr, idx = p.numpy(), i.numpy()
V, E = len(ptr) - 1, len(idx)
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)
B = E * (4 * F + 8) + V * 4 * F + 4 * (V + 1)


def run(ptr, idx, tag):
    agg = gnc.Aggregator_GCN(torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.ones(len(idx), device=dev), F, F)
    out = {"order": tag}
    for name, fn in (("ms", lambda: agg.run(x, y, 512, "balanced")), ("probe_ms", lambda: agg.probe_gather(x, "balanced"))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) / 10
    out["edges_per_s"] = E / (out["ms"] * 1e-3)
    out["gather_frac_of_8TBps"] = B / (out["ms"] * 1e-3) / 8e12
    print(json.dumps(out), flush=True)


run(ptr, idx, "plain (as generated)")
t0 = time.perf_counter()
rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
print(json.dumps({"reorder_generator_s": time.perf_counter() - t0, "host_threads": os.cpu_count(),
                  "walkers": os.environ.get("GNNAGG_REORDER_WALKERS", "auto (64 logical walkers above 20 M edges)")}), flush=True)
assert len(rows) == V and np.array_equal(np.sort(rows), np.arange(V))
nptr, nidx, _ = gnc.reorder_csr(ptr, idx, rows)
run(nptr, nidx, "cache-aware greedy reorder applied on load")
