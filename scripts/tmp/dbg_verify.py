import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
bench.np, bench.torch = np, torch
import gnn_computing_amd as gnc
from oracle import oracle as orc
dev = torch.device("cuda", 0)
# P1
ptr, idx = gnc.graph.dataset("products", device=dev)
V, E, F = ptr.numel() - 1, idx.numel(), 100
val = torch.ones(E, device=dev)
agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
agg.run(x, y, 512, "balanced")
ptr_h = ptr.cpu().numpy()
rows = bench.pick_rows(ptr_h, 200, 1)
sp, si, eids = bench.sample_rows(ptr_h, idx, rows)
xh = x.cpu().numpy()
got = y[torch.from_numpy(rows).to(dev)].cpu().numpy()
chunk, seg = agg.balanced_params()
print("P1 chunk seg parts", chunk, seg, agg.balanced_partitions())
ps, tg = orc.neighbor_grouping(sp, chunk)
ref = orc.gcn_grouped(ps, tg, si, np.ones(len(si), np.float32), xh, len(rows), seg=seg)
bad = np.nonzero((got != ref).any(axis=1))[0]
print("P1 exact-bad rows", [(int(rows[b]), int(sp[b+1]-sp[b])) for b in bad][:20])
sc = orc.gcn_abs_scale(sp, si, np.ones(len(si), np.float32), xh)
d = np.abs(got - orc.gcn_seq(sp, si, np.ones(len(si), np.float32), xh)) / (1e-5 * sc + 1e-30)
print("P1 bound worst", d.max(), "deg of worst", int(np.diff(sp)[np.unravel_index(d.argmax(), d.shape)[0]]))
del agg, x, y, ptr, idx, val
torch.cuda.empty_cache()
# G
ptr, idx = gnc.graph.dataset("reddit", device=dev)
V, E, H, F = ptr.numel() - 1, idx.numel(), 8, 256
D = 32
for scale_att in (1.0, 0.5):
    agg = gnc.Aggregator_GAT(ptr, idx, F, F)
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    att = torch.randn((V, H, 2), device=dev) * scale_att
    agg.run(x, att, y, 128, "balanced", heads=H)
    ptr_h = ptr.cpu().numpy()
    rows = bench.pick_rows(ptr_h, 40, 1)
    sp, si, eids = bench.sample_rows(ptr_h, idx, rows)
    xh = x.cpu().numpy(); n = len(rows)
    got = y[torch.from_numpy(rows).to(dev)].cpu().numpy()
    chunk, seg = agg.balanced_params(); parts = agg.balanced_partitions()
    atth = att.cpu().numpy(); att_mix = atth.copy(); att_mix[:n, :, 0] = atth[rows, :, 0]
    ps, ix, tg, _ = orc.locality_schedule(sp, si, parts, agg.balanced_partition_columns(), ng=chunk)
    ref, _, _ = orc.gat_grouped(ps, tg, ix, att_mix, xh, n, H, seg=0)
    w = orc.gat_att(sp, si, att_mix, H)
    scale = np.zeros((n, F))
    for k in range(n):
        e0, e1 = int(sp[k]), int(sp[k + 1])
        if e1 > e0:
            scale[k] = np.einsum("eh,ehd->hd", w[e0:e1].astype(np.float64), np.abs(xh[si[e0:e1]]).reshape(e1 - e0, H, D)).reshape(F)
    bound = 1e-5 * (scale + np.abs(ref)) + 1e-30
    r1 = np.abs(got.astype(np.float64) - ref) / bound
    r2 = np.abs(got.astype(np.float64) - orc.gat_fused(sp, si, att_mix, xh, H)) / bound
    print("G att x", scale_att, "worst vs grouped", r1.max(), "deg", int(np.diff(sp)[np.unravel_index(r1.argmax(), r1.shape)[0]]),
          "worst vs fused", r2.max(), "deg", int(np.diff(sp)[np.unravel_index(r2.argmax(), r2.shape)[0]]), "nan", np.isnan(got).any())
    del agg
