#!/usr/bin/env python3
"""Round 6: bench.py ran 5 us slower per launch (78.7 vs 73.7 us) when EVERYTHING -- allocations included -- happened under a non-null
torch stream, and at the old speed when only the timed launches did.  Which allocation matters?  One process, the headline workload, every
launch on the same late-created side stream; the variants differ in which tensors were allocated while a side stream was current."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
ptr_t, idx_t = gnc.graph.dataset("arxiv")
ptr, idx = ptr_t.numpy(), idx_t.numpy()
rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
rptr, ridx, _ = gnc.reorder_csr(ptr, idx, rows)
V, E, F = len(ptr) - 1, len(idx), 128
x = np.random.default_rng(123).standard_normal((V, F), dtype=np.float32)
alloc_stream = torch.cuda.Stream()
run_stream = torch.cuda.Stream()


def ours(fn, warm=10, iters=100):
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


def make(what, side):
    import contextlib
    with (torch.cuda.stream(alloc_stream) if what in side else contextlib.nullcontext()):
        if what == "x":
            t = torch.from_numpy(x).to(dev)
        elif what == "y":
            t = torch.empty((V, F), device=dev)
        elif what == "graph":
            t = (torch.from_numpy(rptr).to(dev), torch.from_numpy(ridx).to(dev), torch.ones(E, device=dev))
        torch.cuda.synchronize()
    return t


for side in ([], ["x"], ["y"], ["graph"], ["agg"], ["x", "y", "graph", "agg"], []):
    import contextlib
    dx, dy = make("x", side), make("y", side)
    dp, di, dv = make("graph", side)
    with (torch.cuda.stream(alloc_stream) if "agg" in side else contextlib.nullcontext()):
        agg = gnc.Aggregator_GCN(dp, di, dv, F, F)
        agg.schedule_balanced(0)
        agg.run(dx, dy, 512, "balanced")
        torch.cuda.synchronize()
    with torch.cuda.stream(run_stream):
        us = ours(lambda: agg.run(dx, dy, 512, "balanced"))
    print(json.dumps({"allocated_under_a_side_stream": side, "balanced_us_on_the_run_stream": round(us, 2),
                      "x_ptr": hex(dx.data_ptr()), "y_ptr": hex(dy.data_ptr()), "idx_ptr": hex(di.data_ptr())}), flush=True)
    del agg, dx, dy, dp, di, dv
    torch.cuda.empty_cache()
