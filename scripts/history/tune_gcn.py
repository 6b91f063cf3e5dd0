#!/usr/bin/env python3
"""A/B harness for the GCN aggregation kernel on one GPU: interleaved rounds of kernel variants over
the arxiv-shaped input (un-reordered / RCM-reordered / community order), median microseconds.
Variants are selected through environment knobs read at aggregator creation:
  GNNAGG_XCD_REMAP (0 identity, 1 equal-count XCD ranges, 2 work-balanced XCD ranges)
  GNNAGG_PLAN      (1 balanced plan kernel, 0 items + combine)
(the per-lane metadata loads, the VEC2 geometries and the persistent streaming kernel this harness also compared
 in round 1 lost and were removed)
"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

F = int(os.environ.get("TUNE_F", "128"))
dev = torch.device("cuda", 0)


def graphs():
    ptr, idx = gnc.graph.dataset("arxiv")
    ptr, idx = ptr.numpy(), idx.numpy()
    out = {"plain": (ptr, idx)}
    rows = gnc.graph.locality_order(ptr, idx)
    nptr, nidx, _ = gnc.reorder_csr(ptr, idx, rows)
    out["rcm"] = (nptr, nidx)
    rows, _ = gnc.cluster_reorder(ptr, idx)
    lptr, lidx, _ = gnc.reorder_csr(ptr, idx, rows)
    out["lsh"] = (lptr, lidx)
    V, E = gnc.graph.SHAPES["arxiv"]
    cp, ci = gnc.graph.powerlaw_csr(V, E, seed=123, community_order=True)
    out["community"] = (cp.numpy(), ci.numpy())
    return out


def main():
    gs = graphs()
    V = len(gs["plain"][0]) - 1
    x = torch.randn((V, F), device=dev)
    y = torch.empty((V, F), device=dev)
    configs = []
    modes = os.environ.get("TUNE_MODES", "rows,balanced").split(",")
    remaps = [int(v) for v in os.environ.get("TUNE_REMAP", "0,1,2").split(",")]
    plans = [int(v) for v in os.environ.get("TUNE_PLAN", "0,1").split(",")]
    for mode, remap, plan in itertools.product(modes, remaps, plans):
        if mode != "balanced" and plan != plans[0]:
            continue
        configs.append(dict(mode=mode, remap=remap, plan=plan))
    extra_chunks = [int(c) for c in os.environ.get("TUNE_CHUNKS", "").split(",") if c]
    aggs = {}
    for gname, (ptr, idx) in gs.items():
        dptr, didx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
        dval = torch.ones(len(idx), device=dev)
        for ci, c in enumerate(configs):
            os.environ["GNNAGG_XCD_REMAP"] = str(c["remap"])
            os.environ["GNNAGG_PLAN"] = str(c["plan"])
            a = gnc.Aggregator_GCN(dptr, didx, dval, F, F)
            if c["mode"] == "balanced":
                a.schedule_balanced(0)
            elif c["mode"] == "scheduled":
                a.schedule(gnc.Schedule.neighbor_grouping, [int(os.environ.get("TUNE_NG", "32"))])
            aggs[(gname, ci, 0)] = a
            if c["mode"] == "balanced":
                for ch in extra_chunks:
                    b = gnc.Aggregator_GCN(dptr, didx, dval, F, F)
                    b.schedule_balanced(ch)
                    aggs[(gname, ci, ch)] = b
    times = {k: [] for k in aggs}
    rounds = int(os.environ.get("TUNE_ROUNDS", "7"))
    inner = 20
    for r in range(rounds + 1):
        for k, a in aggs.items():
            mode = configs[k[1]]["mode"]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.run(x, y, 512, mode)
            e0.record()
            for _ in range(inner):
                a.run(x, y, 512, mode)
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[k].append(e0.elapsed_time(e1) * 1e3 / inner)
    print("%-10s %-9s %5s %5s %6s | %9s %9s" % ("graph", "mode", "remap", "plan", "chunk", "med_us", "min_us"))
    for k in sorted(times, key=lambda k: (k[0], np.median(times[k]))):
        c = configs[k[1]]
        print("%-10s %-9s %5d %5d %6d | %9.1f %9.1f" % (k[0], c["mode"], c["remap"], c["plan"], k[2],
                                                        np.median(times[k]), np.min(times[k])))


if __name__ == "__main__":
    main()
