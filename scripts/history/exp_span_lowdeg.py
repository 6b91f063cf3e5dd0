#!/usr/bin/env python3
"""Experiment: the segmented-stream kernel (k_gcn_span) on LOW-degree graphs -- one source range, one 128-float tile on the
caller's X (no re-tiling) -- against the descriptor kernel (k_gcn_plan) of the chunked plan.  arxiv-shaped, greedy reorder."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402
dev = torch.device("cuda", 0)
p, i = gnc.graph.dataset("arxiv"); ptr, idx = p.numpy(), i.numpy()
rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
ptr, idx, _ = gnc.reorder_csr(ptr, idx, rows)
V, F = len(ptr) - 1, 128
x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
dp, di, dv = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.ones(len(idx), device=dev)
ref = None
for opts in ({}, {"partitions": 1, "tile_width": 128, "retile": 0}, {"partitions": 1, "tile_width": 128, "retile": 1},
             {"partitions": 2, "tile_width": 128, "retile": 0}):
    agg = gnc.Aggregator_GCN(dp, di, dv, F, F)
    for k, v in opts.items():
        agg.set_option(k, v)
    for _ in range(20): agg.run(x, y, 512, "balanced")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): agg.run(x, y, 512, "balanced")
    b.record(); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    err = float((y - ref).abs().max())
    print(json.dumps({"opts": opts, "us": a.elapsed_time(b) * 1e3 / 200, "partitions": agg.balanced_partitions(), "chunk": agg.balanced_params()[0], "max_abs_diff_vs_chunked": err}), flush=True)
