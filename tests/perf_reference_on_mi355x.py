#!/usr/bin/env python3
"""The reference's OWN kernels (oracle/_ref/libref_gnn.so: /root/reference translated by hipify-perl, see oracle/ref_build.sh)
timed on this MI355X beside this library, on the headline workload and the reference drivers' default shapes.  Context for
DESIGN.md -- not a bench line and not a pytest (it lives under tests/ because only tests/ may load oracle/): the reference was written for 32-lane warps (two of them share a wavefront here)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import gnn_computing_amd as gnc  # noqa: E402
from oracle import ref  # noqa: E402

dev = torch.device("cuda", 0)


RUN_STREAM = None


def ours(fn, warm=10, iters=50):
    """On ONE NON-NULL stream made for the launches (round 6); inputs stay on the default stream.  Rounds 2-5 timed on
    the null stream, where HIP orders every launch against the process's other streams: after oracle/_ref's kernels and the rows mode
    (which forks two streams) had run, back-to-back null-stream launches of the 74 us balanced kernel cost 79-87 us on the device --
    exactly the locality reorder's gain, which is why this file showed 85.8 vs 86.2 us ("no gain") while bench.py, in a process that had
    only ever used the null stream, showed 85.2 -> 73.5 us (tests/perf_reorder_discrepancy.py host: null stream 74.5 -> 79.4 -> 82.3 us as
    streams appear, a dedicated non-null stream 74.4-74.5 us in every state, host time per call 8 us throughout)."""
    global RUN_STREAM
    if RUN_STREAM is None:
        RUN_STREAM = torch.cuda.Stream()
    RUN_STREAM.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(RUN_STREAM):
        return _ours(fn, warm, iters)


def _ours(fn, warm, iters):
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


def main():
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    rptr, ridx, _ = gnc.reorder_csr(ptr, idx, rows)
    V, E = len(ptr) - 1, len(idx)
    rng = np.random.default_rng(123)
    val = np.ones(E, np.float32)
    att = (rng.standard_normal((V, 2)) * 0.3).astype(np.float32)
    for name, (p, i) in (("no reorder", (ptr, idx)), ("locality reorder on load", (rptr, ridx))):
        dp, di = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
        for F, block in ((128, 512), (32, 512)):
            x = rng.standard_normal((V, F), dtype=np.float32)
            dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)
            agg = gnc.Aggregator_GCN(dp, di, torch.from_numpy(val).to(dev), F, F)
            agg.schedule(gnc.Schedule.neighbor_grouping, [16])
            out = dict(workload="arxiv-shaped %dx%d GCN sum F=%d, %s" % (V, E, F, name),
                       reference_aggr_gcn_us=ref.time_run("gcn", p, i, val, x, block, False),
                       reference_neighbor_grouping16_us=ref.time_run("gcn", p, i, val, x, block, True, 16),
                       reference_neighbor_grouping32_us=ref.time_run("gcn", p, i, val, x, block, True, 32),
                       ours_rows_us=ours(lambda: agg.run(dx, dy, 512, 0)),
                       ours_neighbor_grouping16_us=ours(lambda: agg.run(dx, dy, 512, 1)),
                       ours_balanced_us=ours(lambda: agg.run(dx, dy, 512, "balanced")))
            print(json.dumps(out), flush=True)
        F = 128
        x = rng.standard_normal((V, F), dtype=np.float32)
        dx, dy, datt = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev), torch.from_numpy(att).to(dev)
        gat = gnc.Aggregator_GAT(dp, di, F, F)
        gat.schedule(gnc.Schedule.neighbor_grouping, [32])
        out = dict(workload="arxiv-shaped GAT (1 head) F=128, %s" % name,
                   reference_aggr_gat_us=ref.time_run("gat", p, i, att, x, 128, False),
                   reference_neighbor_grouping32_us=ref.time_run("gat", p, i, att, x, 128, True, 32),
                   ours_rows_us=ours(lambda: gat.run(dx, datt, dy, 128, 0)),
                   ours_neighbor_grouping32_us=ours(lambda: gat.run(dx, datt, dy, 128, 1)),
                   ours_balanced_us=ours(lambda: gat.run(dx, datt, dy, 128, "balanced")))
        print(json.dumps(out), flush=True)


def big():
    """The high-degree / large shapes at a width the reference's launch geometry accepts (feat % 32 == 0): reddit-shaped and
    products-shaped CSR, F = 128, implicit unit weights (the reference always streams a value array)."""
    for name in ("reddit", "products"):
        ptr_t, idx_t = gnc.graph.dataset(name, device=dev)
        p, i = ptr_t.cpu().numpy(), idx_t.cpu().numpy()
        V, E, F = len(p) - 1, len(i), 128
        rng = np.random.default_rng(123)
        x = rng.standard_normal((V, F), dtype=np.float32)
        val = np.ones(E, np.float32)
        dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)
        agg = gnc.Aggregator_GCN(ptr_t, idx_t, torch.from_numpy(val).to(dev), F, F)
        agg.schedule(gnc.Schedule.neighbor_grouping, [32])
        out = dict(workload="%s-shaped %dx%d GCN sum F=%d" % (name, V, E, F),
                   reference_aggr_gcn_us=ref.time_run("gcn", p, i, val, x, 512, False, warm=1, iters=3),
                   reference_neighbor_grouping32_us=ref.time_run("gcn", p, i, val, x, 512, True, 32, warm=2, iters=5),
                   ours_rows_us=ours(lambda: agg.run(dx, dy, 512, 0), 2, 5),
                   ours_neighbor_grouping32_us=ours(lambda: agg.run(dx, dy, 512, 1), 2, 5),
                   ours_balanced_us=ours(lambda: agg.run(dx, dy, 512, "balanced"), 2, 5))
        print(json.dumps(out), flush=True)
        del agg, dx, dy, ptr_t, idx_t
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big()
    else:
        main()
