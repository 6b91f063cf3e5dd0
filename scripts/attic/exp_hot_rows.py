#!/usr/bin/env python3
"""Option "hot_rows" on the reddit-shaped configs: R (SAGE mean, F = 602) and G (GAT 8 x 32) with the hottest rows of every slice
in LDS against the plain streaming form.  usage: exp_hot_rows.py [R|G|RG] [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GNNAGG_HOT_DEBUG", "1")
import gnn_computing_amd as gnc  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "RG"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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


if "R" in which:
    F = 602
    g = torch.Generator(device="cpu").manual_seed(123)
    x = torch.randn(V, F, generator=g).to(dev)
    base = None
    for hot in [int(v) for v in os.environ.get("HOT_LIST", "0,256,128,64").split(",")]:
        agg = gnc.Aggregator_GCN(dptr, didx, None, F, F)
        agg.set_option("hot_rows", hot)
        y = torch.empty(V, F, device=dev)
        t0 = time.time()
        agg.run(x, y, 128, "balanced", reduce="mean")
        torch.cuda.synchronize()
        first = time.time() - t0
        ms = timed(lambda: agg.run(x, y, 128, "balanced", reduce="mean"))
        if base is None:
            base = y.clone()
        print("R  hot_rows %3d: %.3f ms per step (first call %.1f s)  bit_equal_to_plain %s" % (hot, ms, first, bool(torch.equal(y, base))), flush=True)
        del agg
if "G" in which:
    F, H = 256, 8
    g = torch.Generator(device="cpu").manual_seed(123)
    x = torch.randn(V, F, generator=g).to(dev)
    att = (torch.randn(V, H, 2, generator=g) * 0.4).to(dev)
    base = None
    for hot in (0, 256, 128):
        gat = gnc.Aggregator_GAT(dptr, didx, F, F)
        gat.set_option("hot_rows", hot)
        y = torch.empty(V, F, device=dev)
        gat.run(x, att, y, 128, "balanced", heads=H)
        ms = timed(lambda: gat.run(x, att, y, 128, "balanced", heads=H))
        if base is None:
            base = y.clone()
        print("G  hot_rows %3d: %.3f ms per step  bit_equal_to_plain %s" % (hot, ms, bool(torch.equal(y, base))), flush=True)
        del gat
