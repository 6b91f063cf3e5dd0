"""Measured ceilings for row gathers (gnnagg_probe_row_gather), the denominators bench.py quotes its roofline fractions against.

The aggregation kernels move "gather-model" bytes: one feature row (or one column tile of it) per edge.  Most of those bytes are
served by a cache -- an XCD's L2 for the 2-D blocked order, the Infinity Cache when the whole feature matrix fits its 256 MB --
so dividing them by the 8 TB/s HBM figure is not a utilisation.  The probe launch has the kernels' own access shape (coalesced id
loads, lane groups of seg_bytes / 16 lanes, 8 gathers in flight) on ids drawn uniformly from a window of a chosen size, and
nothing else: no graph, no FMA chain, no store.  Known bytes / its time = what the memory system offers to that pattern when the
rows live where the window puts them.  No reference counterpart (the reference reports L2 hit rates from nvprof, Figure9/run.sh).
"""
import ctypes

import torch

from ._lib import check, lib

N_BLOCKS = 256 * 8 * 4          # four rounds of eight 256-thread workgroups per CU
BYTES_PER_GROUP = 2048 * 128    # the same gathered bytes in every case (scripts/micro/gather_ceiling.hip)


def row_gather_ceiling(device, seg_bytes, pitch_bytes, window_bytes, private_per_xcd=False, reps=5, seed=12345):
    """GB/s (useful gathered bytes / median launch time) of seg_bytes-byte row gathers at pitch_bytes, ids uniform over a window of
    window_bytes (cache footprint in 128-byte lines).  private_per_xcd: workgroup b (XCD b % 8) draws from window b % 8 -- eight
    disjoint windows, one per L2 -- instead of all from one.  Returns a dict with the rate, the time and what was measured."""
    active = seg_bytes // 16
    lanes = 8
    while lanes < active:
        lanes *= 2
    gpb = 256 // lanes
    per_group = max(lanes, (BYTES_PER_GROUP // (lanes * 16)) // lanes * lanes)
    foot = seg_bytes if pitch_bytes % 128 == 0 and seg_bytes % 128 == 0 else (seg_bytes + 127) // 128 * 128 + 128
    window_rows = max(1, int(window_bytes // foot))
    n_windows = 8 if private_per_xcd else 1
    rows = torch.ones((n_windows * window_rows * pitch_bytes + 4096) // 4, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ids = torch.randint(0, window_rows, (N_BLOCKS, gpb * per_group), generator=g, device=device, dtype=torch.int32)
    if private_per_xcd:
        ids += (torch.arange(N_BLOCKS, device=device, dtype=torch.int32) % 8 * window_rows)[:, None]
    ids = ids.contiguous()
    n_ids = ids.numel()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        check(lib().gnnagg_probe_row_gather(ctypes.c_void_p(rows.data_ptr()), pitch_bytes, seg_bytes, ctypes.c_void_p(ids.data_ptr()), n_ids,
                                            per_group, stream))
    launch()
    launch()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    useful = n_ids * seg_bytes
    return {"gbps": useful / t / 1e9, "us": t * 1e6, "seg_bytes": seg_bytes, "pitch_bytes": pitch_bytes, "window_bytes": int(window_rows * foot),
            "windows": n_windows, "gathers": n_ids, "useful_bytes": useful,
            "what": "%d-B segments of %d-B rows, ids uniform over %s%.1f MB (128-B-line footprint), %d gathers, median of %d launches of "
                    "gnnagg_probe_row_gather" % (seg_bytes, pitch_bytes, "8 per-XCD windows of " if private_per_xcd else "one window of ",
                                                 window_rows * foot / 1048576.0, n_ids, reps)}
