import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gnn_computing_amd as gnc
from oracle import oracle as orc
dev = torch.device("cuda", 0)
rng = np.random.default_rng(23)
for F in (64, 100, 128, 256, 600):
    for weights in (True, False):
        V = 3000
        deg = rng.integers(0, 10, V)
        deg[[5, 700, 1500, 2999]] = [9001, 1500, 4097, 2240]
        ptr = np.zeros(V + 1, np.int32); ptr[1:] = np.cumsum(deg)
        E = int(ptr[-1])
        idx = rng.integers(0, V, E).astype(np.int32)
        x = rng.standard_normal((V, F)).astype(np.float32)
        val = rng.standard_normal(E).astype(np.float32) if weights else None
        agg = gnc.Aggregator_GCN(torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), None if val is None else torch.from_numpy(val).to(dev), F, F)
        agg.set_option("rows_hub_tile", 2)
        y = torch.full((V, F), 7.0, device=dev)
        oval = val if val is not None else np.ones(E, np.float32)
        for red, fn in (("sum", orc.gcn_seq), ("mean", orc.gcn_mean), ("max", orc.gcn_max)):
            y.fill_(7.0)
            agg.run(torch.from_numpy(x).to(dev), y, 512, 0, reduce=red)
            torch.cuda.synchronize()
            ok = np.array_equal(y.cpu().numpy(), fn(ptr, idx, oval, x))
            print("F", F, "weights", weights, red, "bit-exact" if ok else "MISMATCH", flush=True)
