#!/usr/bin/env python3
"""the canonical rows mode (`scheduled = 0` = aggr_gcn's own order, aggr_gcn.h:5-36) on the headline input, eager
launches against a captured HIP graph.  Eager, the mode forks the hub rows to an auxiliary stream and joins it (two event records, two
stream waits per step); captured, the same launches are kernel nodes without an edge between the two branches.
Prints one JSON line (profiles/r05/rows_mode.txt)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda", 0)
N = int(os.environ.get("ITERS", "200"))


def batch_us(fn, n=N, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / n)
    return float(np.median(best))


def main():
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    V, E, F = len(ptr) - 1, len(idx), 128
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    nptr, nidx, _ = gnc.reorder_csr(ptr, idx, rows)
    x = np.random.default_rng(123).standard_normal((V, F), dtype=np.float32)[rows]
    val = np.ones(E, np.float32)
    dx, y = torch.from_numpy(np.ascontiguousarray(x)).to(dev), torch.empty((V, F), device=dev)
    out = {"config": "A (headline input, reorder applied)", "num_v": V, "num_e": E, "feat": F, "iters_per_batch": N}
    ref = orc.gcn_seq(nptr, nidx, val, x)
    for mode in ("rows", "balanced"):
        agg = gnc.Aggregator_GCN(torch.from_numpy(nptr).to(dev), torch.from_numpy(nidx).to(dev), torch.from_numpy(val).to(dev), F, F)
        step = lambda: agg.run(dx, y, 512, mode)  # noqa: E731
        step()
        torch.cuda.synchronize()
        out[mode + "_eager_us"] = batch_us(step)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        y.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        if mode == "rows":
            assert np.array_equal(y.cpu().numpy(), ref), "captured rows mode differs from the oracle's CSR-order chains"
        out[mode + "_graph_us"] = batch_us(g.replay)
        # ten steps in one graph: what a captured inner loop (a layer stack, an epoch) pays per step
        g10 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g10):
            for _ in range(10):
                step()
        out[mode + "_graph10_us_per_step"] = batch_us(g10.replay, n=max(N // 10, 1)) / 10
        if mode == "rows":
            agg.set_option("aux_stream", 0)
            step()
            torch.cuda.synchronize()
            out["rows_one_stream_eager_us"] = batch_us(step)
            agg.set_option("aux_stream", 1)
    out["rows_verified"] = "captured replay np.array_equal to orc.gcn_seq"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
