#!/usr/bin/env python3
"""rocprofv3 --kernel-trace target (scripts/null_stream_trace.sh): 100 back-to-back launches of the headline kernel on the NULL stream in a
process that has only ever used it, then -- after a second torch stream ran one kernel -- 100 more.  The trace says whether the 5 us the
second phase loses are kernel DURATION or GAPS between kernels."""
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
dx, dy = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
agg = gnc.Aggregator_GCN(torch.from_numpy(rptr).to(dev), torch.from_numpy(ridx).to(dev), torch.ones(E, device=dev), F, F)
agg.schedule_balanced(0)
for phase in range(2):
    for _ in range(120):
        agg.run(dx, dy, 512, "balanced")
    torch.cuda.synchronize()
    if phase == 0:
        with torch.cuda.stream(torch.cuda.Stream()):
            torch.zeros(1 << 20, device=dev).add_(1)     # the marker between the two phases in the trace (an elementwise kernel)
        torch.cuda.synchronize()
