#!/usr/bin/env python3
"""Canonical rows mode (`scheduled = 0`) on the blocked order (option "rows_blocked") against the row kernels, reddit-shaped SAGE mean
F = 602 and GCN sum F = 128 / 256; the balanced order beside them.  usage: exp_rows_blocked.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_computing_amd as gnc  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
ptr, idx = gnc.graph.dataset("reddit")
V, E = ptr.numel() - 1, idx.numel()
dptr, didx = ptr.to(dev), idx.to(dev)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ms[len(ms) // 2]


g = torch.Generator(device="cpu").manual_seed(123)
for F, red in ((602, "mean"), (128, "sum"), (256, "sum")):
    x = torch.randn(V, F, generator=g).to(dev)
    res = {}
    for name, opt, mode in (("rows, blocked chains", 1, 0), ("rows, row kernels", 0, 0), ("balanced", 1, "balanced")):
        agg = gnc.Aggregator_GCN(dptr, didx, None, F, F)
        agg.set_option("rows_blocked", opt)
        y = torch.empty(V, F, device=dev)
        ms = timed(lambda: agg.run(x, y, 128, mode, reduce=red))
        res[name] = y.clone()
        print("F %3d %-4s  %-22s %8.3f ms per step  (ranges %d)" % (F, red, name, ms, agg.rows_blocked_ranges() if mode == 0 else agg.balanced_partitions()), flush=True)
        del agg
    print("      rows: blocked chains == row kernels bit for bit: %s" % bool(torch.equal(res["rows, blocked chains"], res["rows, row kernels"])), flush=True)
