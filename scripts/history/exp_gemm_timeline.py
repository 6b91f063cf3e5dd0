#!/usr/bin/env python3
"""Where a wavefront of the wide GEMM spends a K chunk (A/B build with -DGNNAGG_GEMM_TIMELINE: s_memtime stamps of wave 0 of every workgroup
at six points of the chunk loop).  usage: GNNAGG_LIB=<timeline build> exp_gemm_timeline.py [M K N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (524288, 512, 128)
dev = torch.device("cuda", 0)
A, B = torch.randn((M, K), device=dev), torch.randn((K, N), device=dev)
C = gnc.matmul_NN(A, B)
L = ctypes.CDLL(os.environ["GNNAGG_LIB"])
nwg = 512 * ((N + 127) // 128)
buf = torch.zeros((nwg, 64, 8), dtype=torch.int64, device=dev)
for _ in range(20):
    gnc.matmul_NN(A, B, C)
torch.cuda.synchronize()
assert L.gnnagg_debug_set_gemm_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
gnc.matmul_NN(A, B, C)
torch.cuda.synchronize()
L.gnnagg_debug_set_gemm_timeline(ctypes.c_void_p(0))
t = buf.cpu().numpy().astype(np.float64)
t = t[(t[:, 8, 0] > 0)]                     # workgroups that ran at least 9 chunks
g = slice(4, 60)                            # steady-state chunks
names = ["fetch issue", "LDS reads + 64 MFMAs", "C stores (last chunk of a tile)", "stash (wait loads, LDS writes)", "barrier"]
period = (t[:, 5:61, 0] - t[:, 4:60, 0]).mean()
print("M=%d K=%d N=%d: %d workgroups; chunk period %.0f s_memtime ticks" % (M, K, N, len(t), period))
for i, n in enumerate(names):
    d = t[:, g, i + 1] - t[:, g, i]
    print("  %-34s mean %7.1f  median %7.1f  p90 %7.1f  (%.1f %% of the period)" % (n, d.mean(), np.median(d), np.percentile(d, 90), 100 * d.mean() / period))
d = t[:, 5:61, 0] - t[:, 4:60, 5]
print("  %-34s mean %7.1f" % ("loop back-edge", d.mean()))
