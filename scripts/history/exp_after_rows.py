"""Is the balanced launch slower once the rows mode (auxiliary stream + fork / join events) has run in the process?"""
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
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
print("balanced, fresh process          %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
other = gnc.Aggregator_GCN(ptr, idx, val, F, F)
other.run(x, y, 512, 0); torch.cuda.synchronize()
print("balanced after ANOTHER handle ran rows mode   %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
del other; torch.cuda.synchronize()
print("balanced after that handle is destroyed       %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    z = torch.zeros(10, device=dev) + 1
torch.cuda.synchronize()
print("balanced after a torch side stream was used   %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
agg.run(x, y, 512, 0); torch.cuda.synchronize()
print("balanced after THIS handle ran rows mode      %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
print("rows %.1f us" % t(lambda: agg.run(x, y, 512, 0)))
print("balanced again %.1f us" % t(lambda: agg.run(x, y, 512, "balanced")))
