import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
bench.np, bench.torch = np, torch
import gnn_computing_amd as gnc
from oracle import oracle as orc
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("products", device=dev)
V, E, F = ptr.numel() - 1, idx.numel(), 100
val = torch.ones(E, device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
for trial in range(3):
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    agg.run(x, y, 512, "balanced")
    ptr_h = ptr.cpu().numpy()
    hub = int(np.diff(ptr_h).argmax())
    rows = np.array([hub])
    sp, si, eids = bench.sample_rows(ptr_h, idx, rows)
    xh = x.cpu().numpy()
    got = y[hub].cpu().numpy()
    ones = np.ones(len(si), np.float32)
    seq = orc.gcn_seq(sp, si, ones, xh)[0]
    exact = xh[si].astype(np.float64).sum(axis=0)
    sc = orc.gcn_abs_scale(sp, si, ones, xh)[0]
    print("trial", trial, "deg", len(si), "scale", sc[:3], "|got-exact| max", np.abs(got - exact).max(), "|seq-exact| max", np.abs(seq - exact).max(),
          "ratio got", (np.abs(got - exact) / (1e-5 * sc)).max(), "ratio seq", (np.abs(seq - exact) / (1e-5 * sc)).max(), "exact[:3]", exact[:3])
