#!/usr/bin/env python3
"""Config G (reddit-shaped GAT 8 x 32) in GNNAGG_MODE_ROWS: canonical chains on the 2-D blocked order (k_gat_span<..., CHAIN>, VERDICT r3
item 6) against the row kernels ("rows_blocked" = 0) and the balanced order; results compared."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.dataset("reddit", device=dev)
V, E = ptr.numel() - 1, idx.numel()
for F, H in ((256, 8), (128, 1), (64, 1)):
    x, att = torch.randn((V, F), device=dev), torch.randn((V, H, 2), device=dev) * 0.4
    ys = {}
    for name, opts, mode in (("rows, chains on the blocked order", {}, 0), ("rows, row kernels", {"rows_blocked": 0}, 0), ("balanced", {}, "balanced")):
        g = gnc.Aggregator_GAT(ptr, idx, F, F)
        for k, v in opts.items():
            g.set_option(k, v)
        y = torch.empty((V, F), device=dev)
        g.run(x, att, y, 128, mode, heads=H)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.run(x, att, y, 128, mode, heads=H)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        ys[name] = y
        info = g.plan_info()
        print("GAT %d x %d  %-36s %7.2f ms per step  (ranges %d, chain plan %.3f s)" % (H, F // H, name, ms, g.rows_blocked_ranges() if mode == 0 else g.balanced_partitions(), info["rows_plan_s"]), flush=True)
        del g
    a, b = ys["rows, chains on the blocked order"], ys["rows, row kernels"]
    print("   chains vs row kernels: %s; max |chains - balanced| = %.3g" % ("bit-equal" if torch.equal(a, b) else "max diff %.3g" % float((a - b).abs().max()),
                                                                            float((a - ys["balanced"]).abs().max())), flush=True)
    torch.cuda.empty_cache()
