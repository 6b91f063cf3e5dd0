#!/usr/bin/env python3
"""Figure 10b on the arxiv-shaped input: aggregation + separate GEMM vs run_with_nn (GEMM as the epilogue).
(The back-to-back arm calls run + matmul_nn itself; the GNNAGG_FUSE_NN switch of rounds 1-3 is gone.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def t(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


name = os.environ.get("FNN_DATASET", "arxiv")
V, E = gnc.graph.SHAPES[name][:2]
shapes = [tuple(int(v) for v in t_.split("x")) for t_ in os.environ.get("FNN_SHAPES", "128x32,128x64,64x32,32x32,256x64").split(",")]
for (F, OUT) in shapes:
    ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev)
    val = torch.randn(E, device=dev)
    x, w = torch.randn((V, F), device=dev), torch.randn((F, OUT), device=dev)
    y, tr, tr2 = torch.empty((V, F), device=dev), torch.empty((V, OUT), device=dev), torch.empty((V, OUT), device=dev)
    agg = gnc.Aggregator_GCN(ptr, idx, val, F, OUT)
    agg.schedule_balanced(0)
    us_a = t(lambda: agg.run(x, y, 128, "balanced"))
    us_g = t(lambda: gnc.matmul_NN(y, w, tr2))

    def base():
        agg.run(x, y, 128, "balanced")
        gnc.matmul_NN(y, w, tr2)

    us_b = t(base)
    us_f = t(lambda: agg.run_with_nn(x, y, w, tr, 128, "balanced"))
    print("F=%d OUT=%d: aggregation %.1f us, GEMM %.1f us, back to back %.1f us | run_with_nn %.1f us | equal %s" % (
        F, OUT, us_a, us_g, us_b, us_f, bool(torch.equal(tr, tr2))))
