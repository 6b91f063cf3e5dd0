#!/usr/bin/env python3
"""Times the balanced mode and its gather probe on one config (R: reddit-shaped SAGE mean F=602; G: GAT 8x32) for a list of
handle-option sets: python scripts/tune_blocked.py R "slice_kb=2560" "slice_kb=3584,tile_width=64" ..."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
cfg = sys.argv[1]
sets = sys.argv[2:] or [""]


def timeit(fn, iters=8, warm=2):
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
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


name = {"R": "reddit", "G": "reddit", "P1": "products"}[cfg]
ptr, idx = gnc.graph.dataset(name, device=dev)
V, E = ptr.numel() - 1, idx.numel()
F = {"R": 602, "G": 256, "P1": 100}[cfg]
x = torch.randn((V, F), device=dev)
y = torch.empty((V, F), device=dev)
att = torch.randn((V, 8, 2), device=dev) * 0.3
for opts in sets:
    agg = gnc.Aggregator_GAT(ptr, idx, F, F) if cfg == "G" else gnc.Aggregator_GCN(ptr, idx, None if cfg == "R" else torch.ones(E, device=dev), F, F)
    for kv in filter(None, opts.split(",")):
        k, v = kv.split("=")
        agg.set_option(k, int(v))
    if cfg == "G":
        run = lambda: agg.run(x, att, y, 128, "balanced", heads=8)  # noqa: E731
    else:
        run = lambda: agg.run(x, y, 128, "balanced", reduce="mean" if cfg == "R" else "sum")  # noqa: E731
    t = timeit(run)
    out = {"config": cfg, "opts": opts, "ms": t, "partitions": agg.balanced_partitions(), "chunk": agg.balanced_params()[0]}
    if cfg != "G":
        try:
            out["probe_ms"] = timeit(lambda: agg.probe_gather(x, "balanced"))
        except Exception as e:  # noqa
            out["probe_ms"] = str(e)
    print(json.dumps(out), flush=True)
    del agg
