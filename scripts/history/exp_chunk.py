"""Balanced-mode chunk size on the headline workload (reordered arxiv-shaped, F=128), with the 4-gather plan kernel:
chunk 32: 81.0, 48: 79.6, 64 (library default): 73.7-74.3, 96: 81.1, 128: 79.0, 256: 123.7 us."""
import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
p, i = gnc.graph.dataset("arxiv"); p, i = p.numpy(), i.numpy()
rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
p, i, _ = gnc.reorder_csr(p, i, rows)
ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
V, E, F = len(p) - 1, len(i), 128
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
val = torch.ones(E, device=dev)
def t(fn, warm=20, iters=200):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for chunk in (0, 32, 48, 64, 96, 128, 256):
    agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
    agg.schedule_balanced(chunk)
    print("chunk", chunk, agg.balanced_params(), "%.2f us" % t(lambda: agg.run(x, y, 512, "balanced")))
