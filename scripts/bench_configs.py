#!/usr/bin/env python3
"""Perf of the other BASELINE.json configs on one MI355X (parity-test cases, not bench lines):
 R: reddit-shaped CSR (232 965 x 114 615 891), GraphSAGE mean, feat=602
 G: reddit-shaped, GAT 8 heads x 32 (feat=256), fused edge-softmax + weighted SpMM
 P1: products-shaped CSR (2 449 029 x 123 718 280), GCN sum feat=100 on ONE GPU (the 1-GPU point of config P)
Prints one JSON line per (config, mode) with median microseconds, edges/s, algorithmic GB/s and roofline fraction."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
WHICH = os.environ.get("CONFIGS", "R,G,P1").split(",")
ITERS = int(os.environ.get("ITERS", "10"))


def timeit(fn, iters=ITERS, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.median(ts))


def emit(cfg, mode, sec, V, E, F, nbytes, **kw):
    print(json.dumps(dict(config=cfg, mode=mode, us=sec * 1e6, edges_per_s=E / sec, algorithmic_gbps=nbytes / sec / 1e9,
                          frac_of_8TBps=nbytes / sec / 8e12, V=V, E=E, F=F, **kw)), flush=True)


def graph(name):
    t0 = time.time()
    # ORDER=community numbers the nodes in the generator's hidden community order (what a perfect locality reorder yields)
    V0, E0 = gnc.graph.SHAPES[name]
    ptr, idx = gnc.graph.powerlaw_csr(V0, E0, seed=123, device=dev, community_order=os.environ.get("ORDER", "plain") == "community")
    torch.cuda.synchronize()
    print("# generated %s in %.1fs" % (name, time.time() - t0), file=sys.stderr, flush=True)
    return ptr, idx


def main():
    if "R" in WHICH or "G" in WHICH:
        ptr, idx = graph("reddit")
        V, E = ptr.numel() - 1, idx.numel()
        if "R" in WHICH:
            F = 602
            x = torch.randn((V, F), device=dev)
            y = torch.empty((V, F), device=dev)
            agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
            nbytes = E * (4 * F + 4) + V * 4 * F + 4 * (V + 1)  # implicit weights: no val stream
            for mode in os.environ.get("MODES", "balanced,rows").split(","):
                emit("R reddit SAGE mean F=602", mode, timeit(lambda: agg.run(x, y, 512, mode, reduce="mean")), V, E, F, nbytes,
                     source_partitions=agg.balanced_partitions() if mode == "balanced" else 0)
            del agg, x, y
        if "G" in WHICH:
            H, D = 8, 32
            F = H * D
            x = torch.randn((V, F), device=dev)
            att = torch.randn((V, H, 2), device=dev)
            y = torch.empty((V, F), device=dev)
            gat = gnc.Aggregator_GAT(ptr, idx, F, F)
            nbytes = E * (4 * F + 4 + 4 * H) + V * (4 * F + 4 * H) + 4 * (V + 1)
            for mode in os.environ.get("MODES", "balanced,rows").split(","):
                emit("G reddit GAT 8x32", mode, timeit(lambda: gat.run(x, att, y, 128, mode, heads=H)), V, E, F, nbytes,
                     source_partitions=gat.balanced_partitions() if mode == "balanced" else 0)
            del gat, x, y, att
        del ptr, idx
        torch.cuda.empty_cache()
    if "P1" in WHICH:
        ptr, idx = graph("products")
        V, E = ptr.numel() - 1, idx.numel()
        F = 100
        x = torch.randn((V, F), device=dev)
        y = torch.empty((V, F), device=dev)
        val = torch.ones(E, device=dev)
        agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
        nbytes = E * (4 * F + 8) + V * 4 * F + 4 * (V + 1)
        for mode in os.environ.get("MODES", "balanced,rows").split(","):
            emit("P1 products GCN F=100 (1 GPU)", mode, timeit(lambda: agg.run(x, y, 512, mode)), V, E, F, nbytes,
                 source_partitions=agg.balanced_partitions() if mode == "balanced" else 0)


if __name__ == "__main__":
    main()
