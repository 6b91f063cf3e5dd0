import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gnn_computing_amd as gnc
from oracle import oracle as orc
dev = torch.device("cuda", 0)
for (M, N, K) in [(70536, 128, 40), (131072, 128, 8), (131072, 128, 32), (131072, 128, 64), (131072, 128, 96), (16384, 128, 8)]:
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K), dtype=np.float32); B = rng.standard_normal((K, N), dtype=np.float32)
    C = gnc.matmul_NN(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)).cpu().numpy()
    ref = orc.matmul_nn(A, B)
    bad = np.argwhere(C != ref)
    print(M, N, K, "mismatches", len(bad), "rows", (np.unique(bad[:, 0])[:12].tolist() if len(bad) else []), "cols", (np.unique(bad[:, 1])[:12].tolist() if len(bad) else []),
          "n bad rows", len(np.unique(bad[:, 0])) if len(bad) else 0)
    if len(bad):
        r, c = bad[0]
        print("   first", r, c, C[r, c], ref[r, c], "row%128", r % 128, "is C zero?", C[r, c] == 0)
