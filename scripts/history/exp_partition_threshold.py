#!/usr/bin/env python3
"""Where the 2-D blocked balanced mode starts to pay: V rows (argv[1], default 400 k), average degree swept, chunked plan
("partitions" = 0) against the blocked order with the library's range count (forced on: "partition_min_degree" = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def t(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


V = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
for deg in (32, 64, 100, 150, 200, 300):
    for F in (128, 256):
        ptr, idx = gnc.graph.powerlaw_csr(V, V * deg, seed=123, device=dev)
        x = torch.randn((V, F), device=dev)
        y = torch.empty((V, F), device=dev)
        res = []
        for opts in ({"partitions": 0}, {"partition_min_degree": 1}):
            agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
            for k, v in opts.items():
                agg.set_option(k, v)
            agg.schedule_balanced(0)
            ms = t(lambda: agg.run(x, y, 128, "balanced"))
            res.append("P=%d %.2f ms" % (agg.balanced_partitions(), ms))
            del agg
        print("V %d deg %d F=%d: chunked %s | blocked %s" % (V, deg, F, res[0], res[1]), flush=True)
        del x, y, ptr, idx
