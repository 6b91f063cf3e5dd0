#!/usr/bin/env python3
"""Where the source-partitioned balanced mode starts to pay: V = 400 k rows, average degree swept, chunked plan vs 16 ranges.
Run twice: GNNAGG_PARTITIONS=0 and GNNAGG_PARTITIONS=16 (the knob is read once per process)."""
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


V = 400000
for deg in (64, 100, 150, 200, 300):
    for F in (128, 256):
        ptr, idx = gnc.graph.powerlaw_csr(V, V * deg, seed=123, device=dev)
        x = torch.randn((V, F), device=dev)
        y = torch.empty((V, F), device=dev)
        agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
        agg.schedule_balanced(0)
        ms = t(lambda: agg.run(x, y, 128, "balanced"))
        print("deg %d F=%d partitions=%d: %.2f ms" % (deg, F, agg.balanced_partitions(), ms), flush=True)
        del agg, x, y, ptr, idx
