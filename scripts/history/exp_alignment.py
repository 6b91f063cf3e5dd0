"""Does the placement of X / Y in device memory change the headline kernel's time?  (A later allocation in a long-lived process
measured 86 us against 74 us for the first ones.)"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
p, i = gnc.graph.dataset("arxiv"); p, i = p.numpy(), i.numpy()
rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
p, i, _ = gnc.reorder_csr(p, i, rows)
ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
V, E, F = len(p) - 1, len(i), 128
def t(fn, warm=10, iters=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
val = torch.ones(E, device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
n = V * F
big = torch.empty(n * 2 + (64 << 20), dtype=torch.float32, device=dev)
print("big base %x" % big.data_ptr())
src = torch.randn(n, device=dev)
for xoff_b, yoff_b in ((0, 0), (4096, 0), (65536, 0), (1 << 20, 0), (0, 4096), (0, 1 << 20), (512, 0), (128 * 3, 0), ((2 << 20) - 512, 0), (0, 0)):
    xo = xoff_b // 4
    x = big[xo:xo + n].view(V, F); x.copy_(src.view(V, F))
    ybase = n + (16 << 20) // 4 + yoff_b // 4
    y = big[ybase:ybase + n].view(V, F)
    print("x off %8d  y off %8d  x %% 2MB = %7d  y %% 2MB = %7d : %.1f us" % (xoff_b, yoff_b, x.data_ptr() % (2 << 20), y.data_ptr() % (2 << 20), t(lambda: agg.run(x, y, 512, "balanced"))))
# separate allocations, as callers make them
for k in range(6):
    x = torch.randn((V, F), device=dev); y = torch.empty((V, F), device=dev)
    print("separate alloc %d: x %x y %x  (x %% 2MB %d, y %% 2MB %d): %.1f us" % (k, x.data_ptr(), y.data_ptr(), x.data_ptr() % (2 << 20), y.data_ptr() % (2 << 20), t(lambda: agg.run(x, y, 512, "balanced"))))
    keep = torch.empty((k + 1) * 12345677, device=dev)  # perturb the allocator
