#!/usr/bin/env python3
"""Times the dense combine GEMM (gnnagg_matmul_nn) at the shapes of the 3-layer model and checks it against torch.mm."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

dev = torch.device("cuda", 0)


def t(fn, it=20):
    """median of 5 timed bursts of `it` calls behind ~60 ms of the same work: a burst right after an idle period runs at ramping clocks
    (round 3's numbers for the first shape of a run -- 248 to 275 us for the 512 -> 128 layer -- carried that)"""
    import time
    t_end = time.perf_counter() + (0.06 if t.warm else 0.5)   # (the first shape of a process: half a second)
    t.warm = True
    while time.perf_counter() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / it)
    return sorted(ts)[len(ts) // 2]


t.warm = False
shapes = [(169343, 128, 32), (169343, 128, 64), (169343, 512, 128), (232965, 602, 128), (2449029, 100, 32)]
if os.environ.get("GEMM_SHAPES"):
    shapes = [tuple(int(v) for v in t_.split("x")) for t_ in os.environ["GEMM_SHAPES"].split(",")]
for (M, K, N) in shapes:
    A, B = torch.randn((M, K), device=dev), torch.randn((K, N), device=dev)
    C = gnc.matmul_NN(A, B)
    ref = A @ B
    err = float((C - ref).abs().max() / ref.abs().max())
    us, us_t = t(lambda: gnc.matmul_NN(A, B, C)), t(lambda: torch.mm(A, B, out=ref))
    byts = 4.0 * (M * K + K * N + M * N)
    print("M=%d K=%d N=%d: %.1f us (%.0f GB/s, %.1f TFLOP/s) | torch.mm %.1f us | max rel diff %.1e" % (
        M, K, N, us, byts / us / 1e3, 2.0 * M * N * K / us / 1e6, us_t, err))
