"""Hunt for intermittent stalls: per-launch GPU (event) and host (perf_counter) times of every GCN mode on the arxiv-shaped
input.  Finding (MI355X boxes of this pool): GPU times are steady; the HOST is descheduled for 6.8 / 15.7 / 45-55 ms in ~0.4 % of
the iterations, with the Python collector on or off (NOGC=1) -- CPU-quota throttling of the container, not the library."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("arxiv", device=dev)
V, E = ptr.numel() - 1, idx.numel()
val = torch.ones(E, device=dev)
bad = 0
import gc
if os.environ.get("NOGC") == "1":
    gc.disable()
for F in (32, 128):
    for rep in range(6):
        x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
        agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
        agg.schedule(gnc.Schedule.neighbor_grouping, [32])
        for name, fn in (("rows", lambda: agg.run(x, y, 512, 0)), ("ng32", lambda: agg.run(x, y, 512, 1)), ("bal", lambda: agg.run(x, y, 512, "balanced")),
                         ("mean", lambda: agg.run(x, y, 512, "balanced", reduce="mean")), ("max", lambda: agg.run(x, y, 512, "balanced", reduce="max"))):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append((a.elapsed_time(b) * 1e3, (time.perf_counter() - t0) * 1e6))
            g = np.array([t[0] for t in ts]); h = np.array([t[1] for t in ts])
            if g.max() > 5 * np.median(g) or h.max() > 20 * np.median(h):
                bad += 1
                print("F=%d rep %d %s: gpu median %.1f max %.1f (iter %d) | host median %.1f max %.1f (iter %d)" % (F, rep, name, np.median(g), g.max(), g.argmax(), np.median(h), h.max(), h.argmax()))
print("outliers:", bad)
