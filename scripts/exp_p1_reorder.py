#!/usr/bin/env python3
"""P1 (products-shaped GCN F=100, one GPU) with the locality reorder applied on load.  The cache-aware greedy order of this
124 M-edge graph takes 5.4 minutes of one CPU core (gnnagg_cluster_reorder_ex, order_mode 1, cluster_cap 1, 8192 cache rows),
so it is computed off-line and passed in like a <dset>.reorder_thres_0.2 file:
    python -c "import numpy as np, gnn_computing_amd as g; p, i = g.graph.dataset('products'); \\
               np.save('scripts/cache/products_greedy_rows.npy', g.cluster_reorder(p.numpy(), i.numpy(), order='cache_greedy', cluster_cap=1, cache_rows=8192)[0])"
The graph is generated on the CPU here (the CPU and GPU generators of torch draw different graphs for the same seed)."""
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
rows = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "products_greedy_rows.npy"))
assert len(rows) == V
nptr, nidx, _ = gnc.reorder_csr(ptr, idx, rows)
run(nptr, nidx, "cache-aware greedy reorder applied on load")
