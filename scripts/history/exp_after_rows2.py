"""Which part of the rows mode leaves the balanced launch slower afterwards?  One scenario per process (argv[1])."""
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
what = sys.argv[1]
if what.endswith("@side"):   # everything on a non-default torch stream
    what = what[:-5]
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
base = t(lambda: agg.run(x, y, 512, "balanced"))
if what == "rows_small":      # rows mode on a graph without long rows: no auxiliary stream, no long-row kernel
    sp, si = gnc.graph.uniform_random_csr(5000, 40000, 3)
    o = gnc.Aggregator_GCN(torch.from_numpy(sp).to(dev), torch.from_numpy(si).to(dev), None, F, F)
    xs, ys = torch.randn((5000, F), device=dev), torch.empty((5000, F), device=dev)
    o.run(xs, ys, 512, 0)
elif what == "rows_other":
    o = gnc.Aggregator_GCN(ptr, idx, val, F, F); o.run(x, y, 512, 0)
elif what == "rows_self":
    agg.run(x, y, 512, 0)
elif what == "mean_self":
    agg.run(x, y, 512, "balanced", reduce="mean")
elif what == "torch_fork_join":   # no library stream at all: a fork / join between torch's current stream and a side stream
    main = torch.cuda.current_stream()
    s2 = torch.cuda.Stream()
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(main); s2.wait_event(e1)
    with torch.cuda.stream(s2):
        z = torch.zeros(1000, device=dev) + 1
    e2.record(s2); main.wait_event(e2)
elif what == "torch_side_only":
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        z = torch.zeros(1000, device=dev) + 1
elif what == "gat":
    g = gnc.Aggregator_GAT(ptr, idx, F, F); g.run(x, torch.randn((V, 2), device=dev), y, 128, "balanced")
torch.cuda.synchronize()
print("%-12s before %.1f us   after %.1f us" % (what, base, t(lambda: agg.run(x, y, 512, "balanced"))))
