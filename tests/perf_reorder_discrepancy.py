#!/usr/bin/env python3
"""tests/perf_reference_on_mi355x.py reported NO gain from the locality reorder for this library's balanced mode
(85.8 vs 86.2 us) while bench.py / rocprofv3 / drivers/fig9 show 85.2 -> 73.5 us on the same input.  This script times the balanced
launch on both numberings under every difference between the two harnesses, one at a time, to find which one eats the gain.
(Lives under tests/ because one factor loads oracle/_ref.)  Output: one JSON line per case."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def ours(fn, warm=10, iters=50):
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
    V, E, F = len(ptr) - 1, len(idx), 128
    rng = np.random.default_rng(123)
    val = np.ones(E, np.float32)
    x = rng.standard_normal((V, F), dtype=np.float32)
    arms = (("no reorder", ptr, idx), ("reorder", rptr, ridx))
    use_ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
    if use_ref:
        from oracle import ref  # noqa: E402

    def case(name, prepare):
        out = {"case": name}
        for arm, p, i in arms:
            dp, di = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
            dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)
            agg = gnc.Aggregator_GCN(dp, di, torch.from_numpy(val).to(dev), F, F)
            prepare(agg, dx, dy, p, i)
            out[arm + "_us"] = round(ours(lambda: agg.run(dx, dy, 512, "balanced")), 2)
            out[arm + "_chunk"] = agg.balanced_params()[0]
        print(json.dumps(out), flush=True)

    case("bench.py's way: schedule_balanced(0), nothing else", lambda a, dx, dy, p, i: a.schedule_balanced(0))
    case("plan built lazily by the first balanced run", lambda a, dx, dy, p, i: None)
    case("after schedule(neighbor_grouping, [16])", lambda a, dx, dy, p, i: a.schedule(gnc.Schedule.neighbor_grouping, [16]))

    def harness(a, dx, dy, p, i):
        a.schedule(gnc.Schedule.neighbor_grouping, [16])
        ours(lambda: a.run(dx, dy, 512, 0))
        ours(lambda: a.run(dx, dy, 512, 1))
    case("the harness's sequence: schedule(16), 60 rows-mode runs, 60 scheduled runs, then balanced", harness)

    def rows_first(a, dx, dy, p, i):
        ours(lambda: a.run(dx, dy, 512, 0))
    case("60 rows-mode runs first (forked streams), then balanced", rows_first)
    if use_ref:
        def with_ref(a, dx, dy, p, i):
            ref.time_run("gcn", p, i, val, x, 512, True, 16)
        case("after the hipified reference's kernels ran in this process", with_ref)


if __name__ == "__main__":
    main()
