#!/usr/bin/env python3
"""Destination-stationary form of the 2-D blocked order (option "dest_stationary", agg_ds.hip) against the streaming form
(k_gcn_span + k_combine_groups) on the reddit-shaped SAGE mean F = 602 workload (config R) -- VERDICT r2 item 1.
usage: exp_dest_stationary.py [feat]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 602
ptr, idx = gnc.graph.dataset("reddit", device=dev)
V, E = ptr.numel() - 1, idx.numel()
x, y, y0 = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev), torch.empty((V, F), device=dev)


def t(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


base = gnc.Aggregator_GCN(ptr, idx, None, F, F)
base.run(x, y0, 512, "balanced", reduce="mean")
print(json.dumps({"form": "streaming (default)", "slice_kb": 4096, "ranges": base.balanced_partitions(),
                  "ms": t(lambda: base.run(x, y0, 512, "balanced", reduce="mean"))}), flush=True)
for slice_kb in (4096, 2048, 1024):
    st = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    st.set_option("slice_kb", slice_kb)
    st.run(x, y0, 512, "balanced", reduce="mean")
    ms_st = t(lambda: st.run(x, y0, 512, "balanced", reduce="mean")) if slice_kb != 4096 else None
    for slack in (0, 1):
        ds = gnc.Aggregator_GCN(ptr, idx, None, F, F)
        ds.set_option("slice_kb", slice_kb)
        ds.set_option("dest_stationary", 1)
        ds.set_option("ds_slack", slack)
        t0 = time.perf_counter()
        ds.run(x, y, 512, "balanced", reduce="mean")
        torch.cuda.synchronize()
        prep = time.perf_counter() - t0
        same = bool(torch.equal(y, y0))
        print(json.dumps({"form": "destination-stationary", "slice_kb": slice_kb, "ranges": ds.balanced_partitions(), "slack": slack,
                          "ms": t(lambda: ds.run(x, y, 512, "balanced", reduce="mean")), "streaming_same_slices_ms": ms_st,
                          "bit_equal_to_streaming": same, "first_run_s": prep}), flush=True)
        del ds
    del st
    torch.cuda.empty_cache()
