import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
p, i = gnc.graph.dataset("arxiv"); p, i = p.numpy(), i.numpy()
rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
p, i, _ = gnc.reorder_csr(p, i, rows)
ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
V, E, F = len(p) - 1, len(i), 128
rng = np.random.default_rng(123)
x_np = rng.standard_normal((V, F), dtype=np.float32)
def t(fn, warm, iters):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for name, x in (("randn", torch.randn((V, F), device=dev)), ("numpy normal", torch.from_numpy(x_np).to(dev))):
    y = torch.empty((V, F), device=dev)
    for vname, val in (("torch.ones", torch.ones(E, device=dev)), ("from numpy", torch.from_numpy(np.ones(E, np.float32)).to(dev))):
        agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
        print(name, vname, [round(t(lambda: agg.run(x, y, 512, "balanced"), w, n), 1) for w, n in ((10, 50), (20, 200), (20, 1000), (10, 50))])
x = torch.from_numpy(x_np).to(dev); y = torch.empty((V, F), device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, torch.ones(E, device=dev), F, F)
agg.schedule(gnc.Schedule.neighbor_grouping, [16])
print("after schedule(ng16): rows", round(t(lambda: agg.run(x, y, 512, 0), 10, 50), 1), "ng16", round(t(lambda: agg.run(x, y, 512, 1), 10, 50), 1),
      "balanced", round(t(lambda: agg.run(x, y, 512, "balanced"), 10, 50), 1), agg.balanced_params())
agg2 = gnc.Aggregator_GCN(ptr, idx, torch.ones(E, device=dev), F, F)
print("fresh handle: rows first", round(t(lambda: agg2.run(x, y, 512, 0), 10, 50), 1), "balanced", round(t(lambda: agg2.run(x, y, 512, "balanced"), 10, 50), 1))
agg3 = gnc.Aggregator_GCN(ptr, idx, torch.ones(E, device=dev), F, F)
print("fresh handle: balanced only", round(t(lambda: agg3.run(x, y, 512, "balanced"), 10, 50), 1))
