#!/usr/bin/env python3
"""Option "overlap_combine" = N >= 2 on the reddit-shaped configs: the ordered combine of a chunk of column tiles on the auxiliary
stream beside the span kernel of the next chunk (N launches instead of one per tile).  usage: exp_overlap_chunks.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_computing_amd as gnc  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 7
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
x = torch.randn(V, 602, generator=g).to(dev)
base = None
for n in (0, 2, 3, 5, 1):
    agg = gnc.Aggregator_GCN(dptr, didx, None, 602, 602)
    agg.set_option("overlap_combine", n)
    y = torch.empty(V, 602, device=dev)
    ms = timed(lambda: agg.run(x, y, 128, "balanced", reduce="mean"))
    if base is None:
        base = y.clone()
    print("R  overlap_combine %d: %.3f ms per step  bit_equal %s" % (n, ms, bool(torch.equal(y, base))), flush=True)
    del agg
x = torch.randn(V, 256, generator=g).to(dev)
att = (torch.randn(V, 8, 2, generator=g) * 0.4).to(dev)
base = None
for n in (0, 2, 4, 1):
    gat = gnc.Aggregator_GAT(dptr, didx, 256, 256)
    gat.set_option("overlap_combine", n)
    y = torch.empty(V, 256, device=dev)
    ms = timed(lambda: gat.run(x, att, y, 128, "balanced", heads=8))
    if base is None:
        base = y.clone()
    print("G  overlap_combine %d: %.3f ms per step  bit_equal %s" % (n, ms, bool(torch.equal(y, base))), flush=True)
    del gat
