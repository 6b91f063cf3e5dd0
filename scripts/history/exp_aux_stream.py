import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
def t(fn, warm=10, iters=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for name, F in (("arxiv", 128), ("arxiv", 32), ("products", 100), ("reddit", 128)):
    ptr, idx = gnc.graph.dataset(name, device=dev)
    V, E = ptr.numel() - 1, idx.numel()
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    print("%s F=%d rows mode (GNNAGG_AUX_STREAM=%s): %.1f us" % (name, F, os.environ.get("GNNAGG_AUX_STREAM", "1"), t(lambda: agg.run(x, y, 512, 0), 3, 20)))
    del agg, x, y, ptr, idx
    torch.cuda.empty_cache()
