#!/usr/bin/env python3
"""Times the single-head GAT forward (with newval) and run_bwd on the arxiv-shaped input."""
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


V, E = gnc.graph.SHAPES["arxiv"]
for F in (128, 32):
    ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=123, device=dev)
    x, att, g = torch.randn((V, F), device=dev), torch.randn((V, 1, 2), device=dev) * 0.5, torch.randn((V, F), device=dev)
    gat = gnc.Aggregator_GAT(ptr, idx, F, F)
    gat.schedule_balanced(0)
    out, newval, div = torch.empty((V, F), device=dev), torch.empty((E, 1), device=dev), torch.empty(V, device=dev)
    d_a_b, d_feat = torch.empty((V, 2), device=dev), torch.empty((V, F), device=dev)
    fwd = t(lambda: gat.run(x, att, out, 128, "balanced", heads=1, newval=newval))
    ctr = t(lambda: gat.run_add_to_center(newval, div))
    bwd = t(lambda: gat.run_bwd(out, g, newval, div, x, d_a_b, d_feat, 0.2))
    print("F=%d: forward (with newval) %.1f us, add_to_center %.1f us, run_bwd %.1f us" % (F, fwd, ctr, bwd))
