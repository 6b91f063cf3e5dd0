#!/usr/bin/env python3
"""Rows mode (`scheduled = 0`, canonical CSR-order chains) timings, and a balanced launch before / after a rows launch in the same
process (the auxiliary stream the rows mode opens for its hub rows costs every later launch ~4 us).  Written for the round-3
experiment that made the hub rows a role of the short-row launch (scripts/attic/rows_long_merged_role.cuh, measured slower and not
kept: profiles/r03/rows_mode_merged_role.txt); profiles/r03/rows_mode.txt is the library as it is.  usage: exp_rows_mode.py [A R G P1 ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def t(fn, warm=5, iters=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for cfg in (sys.argv[1:] or ["A", "P1", "R", "G"]):
    name, F = {"A": ("arxiv", 128), "R": ("reddit", 602), "G": ("reddit", 256), "P1": ("products", 100)}[cfg]
    if cfg == "A":   # the headline input: locality reorder applied on load
        p, i = gnc.graph.dataset(name)
        p, i = p.numpy(), i.numpy()
        rows, _ = gnc.cluster_reorder(p, i, order="cache_greedy", cluster_cap=1, cache_rows=8192)
        p, i, _ = gnc.reorder_csr(p, i, rows)
        ptr, idx = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
    else:
        ptr, idx = gnc.graph.dataset(name, device=dev)
    V, E = ptr.numel() - 1, idx.numel()
    x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
    out = {"config": cfg, "num_v": V, "num_e": E, "feat": F}
    if cfg == "G":
        att = torch.randn((V, 8, 2), device=dev) * 0.3
        agg = gnc.Aggregator_GAT(ptr, idx, F, F)
        bal = lambda: agg.run(x, att, y, 128, "balanced", heads=8)   # noqa: E731
        rows_fn = lambda: agg.run(x, att, y, 128, 0, heads=8)        # noqa: E731
    else:
        agg = gnc.Aggregator_GCN(ptr, idx, torch.ones(E, device=dev) if cfg in ("A", "P1") else None, F, F)
        red = "mean" if cfg == "R" else "sum"
        bal = lambda: agg.run(x, y, 512, "balanced", reduce=red)     # noqa: E731
        rows_fn = lambda: agg.run(x, y, 512, 0, reduce=red)          # noqa: E731
    n = 100 if cfg == "A" else 10
    if os.environ.get("ROWS_AUX"):   # 0: hub rows on the same stream, one launch after the other (per-kernel times stand alone)
        agg.set_option("aux_stream", int(os.environ["ROWS_AUX"]))
    if os.environ.get("ROWS_HUB_TILE") and cfg != "G":
        agg.set_option("rows_hub_tile", int(os.environ["ROWS_HUB_TILE"]))
    if os.environ.get("ROWS_MEDIUM_SET"):
        agg.set_option("rows_medium_edges", int(os.environ["ROWS_MEDIUM_SET"]))
    out["balanced_before_us"] = t(bal, iters=n)
    out["rows_us"] = t(rows_fn, iters=n)
    out["balanced_after_us"] = t(bal, iters=n)
    # ROWS_MEDIUM="-1,64,128,256": the rows mode again per threshold of the medium class ("rows_medium_edges"; -1 = no medium class)
    for m in [int(v) for v in os.environ.get("ROWS_MEDIUM", "").split(",") if v]:
        agg.set_option("rows_medium_edges", m)
        out["rows_medium_%d_us" % m] = t(rows_fn, iters=n)
    print(json.dumps(out), flush=True)
    del agg, x, y
    torch.cuda.empty_cache()
