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


def xcd_mapping():
    """Second question: does the blockIdx -> XCD assignment the kernels' range mapping assumes (workgroup b runs on XCD b % 8) still hold
    after the process has run the rows mode (forked streams -> more hardware queues)?  The instrumented launch (run_clock: s_memrealtime +
    __smid per workgroup) answers it: share of workgroups whose XCC id equals (b + k) % 8 for the best constant k, before and after."""
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    rptr, ridx, _ = gnc.reorder_csr(ptr, idx, rows)
    V, E, F = len(ptr) - 1, len(idx), 128
    x = np.random.default_rng(123).standard_normal((V, F), dtype=np.float32)
    dp, di = torch.from_numpy(rptr).to(dev), torch.from_numpy(ridx).to(dev)
    dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)

    def mapping(agg, tag):
        t = agg.run_clock(dx, dy, 64, 0).cpu().numpy()
        smid = t[:, 2].astype(np.int64)
        b = np.arange(len(smid))
        best = {}
        for shift in range(4, 12):
            xcc = (smid >> shift) & 7
            if len(np.unique(xcc)) < 8:
                continue
            best[shift] = max(float(np.mean(xcc == (b + k) % 8)) for k in range(8))
        print(json.dumps({"state": tag, "workgroups": int(len(smid)), "distinct_smid": int(len(np.unique(smid))),
                          "share_on_xcd_b_mod_8_by_shift": best}), flush=True)

    agg = gnc.Aggregator_GCN(dp, di, torch.ones(E, device=dev), F, F)
    agg.schedule_balanced(0)
    print(json.dumps({"state": "fresh process", "balanced_us": round(ours(lambda: agg.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    mapping(agg, "fresh process")
    print(json.dumps({"state": "after run_clock", "balanced_us": round(ours(lambda: agg.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize()
    print(json.dumps({"state": "after a second torch stream ran a kernel", "balanced_us": round(ours(lambda: agg.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    ours(lambda: agg.run(dx, dy, 512, 0))
    print(json.dumps({"state": "after 60 rows-mode runs", "balanced_us": round(ours(lambda: agg.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    mapping(agg, "after 60 rows-mode runs")
    agg2 = gnc.Aggregator_GCN(dp, di, torch.ones(E, device=dev), F, F)
    agg2.schedule_balanced(0)
    print(json.dumps({"state": "a NEW handle after the rows-mode runs", "balanced_us": round(ours(lambda: agg2.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    torch.cuda.synchronize()
    import time
    time.sleep(2.0)
    print(json.dumps({"state": "after 2 s of idling", "balanced_us": round(ours(lambda: agg2.run(dx, dy, 512, "balanced")), 2)}), flush=True)
    for it in (200, 1000):
        print(json.dumps({"state": "%d back-to-back launches" % it, "balanced_us": round(ours(lambda: agg2.run(dx, dy, 512, "balanced"), 10, it), 2)}), flush=True)


def host_or_device():
    """Third question: is the slow state a slower KERNEL or a slower LAUNCH?  Per state: host seconds per call (no synchronise inside the
    loop), device seconds per launch from one event pair around 200 launches, the kernel alone from a captured HIP graph of 20 launches
    (no host between them), and the same launches issued on a non-null stream."""
    import time
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    rptr, ridx, _ = gnc.reorder_csr(ptr, idx, rows)
    V, E, F = len(ptr) - 1, len(idx), 128
    x = np.random.default_rng(123).standard_normal((V, F), dtype=np.float32)
    dp, di = torch.from_numpy(rptr).to(dev), torch.from_numpy(ridx).to(dev)
    dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)
    agg = gnc.Aggregator_GCN(dp, di, torch.ones(E, device=dev), F, F)
    agg.schedule_balanced(0)
    fn = lambda: agg.run(dx, dy, 512, "balanced")  # noqa: E731

    def host_us(n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        return round(t, 2)

    def graph_us():
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(20):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                g.replay()
            b.record()
            torch.cuda.synchronize()
        return round(a.elapsed_time(b) * 1e3 / 100, 2)

    def side_stream_us():
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            return round(ours(fn, 10, 200), 2)

    def state(tag, with_graph=True):
        out = {"state": tag, "null_stream_device_us": round(ours(fn, 10, 200), 2), "null_stream_host_us_per_call": host_us()}
        print(json.dumps(out), flush=True)

    state("fresh process: only the null stream has ever been used")
    out = {"state": "same process, launches on a NON-null stream", "side_stream_device_us": side_stream_us()}
    print(json.dumps(out), flush=True)
    state("after that side stream existed")
    print(json.dumps({"state": "captured HIP graph of 20 launches (no host between them)", "graph_device_us": graph_us()}), flush=True)
    ours(lambda: agg.run(dx, dy, 512, 0))
    state("after 60 rows-mode runs (the library forks two more streams)")
    print(json.dumps({"state": "after the rows-mode runs: NON-null stream", "side_stream_device_us": side_stream_us()}), flush=True)
    print(json.dumps({"state": "after the rows-mode runs: captured graph", "graph_device_us": graph_us()}), flush=True)


def l2_between_launches():
    """Fourth question: is what a multi-queue process loses the L2 CONTENT one launch leaves for the next?  (With several hardware queues
    the runtime may fence every dispatch at a wider scope -- an L2 invalidate at kernel start.)  Per-launch time of the headline kernel,
    one event pair per launch, (a) launches back to back and (b) with a kernel between them that reads 48 MB of other memory -- more than
    the 32 MB of L2, far less than the Infinity Cache -- for the reordered graph, the original numbering and uniform-random ids."""
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    rows, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
    rptr, ridx, _ = gnc.reorder_csr(ptr, idx, rows)
    V, E, F = len(ptr) - 1, len(idx), 128
    x = np.random.default_rng(123).standard_normal((V, F), dtype=np.float32)
    uid = np.random.default_rng(7).integers(0, V, E).astype(np.int32)
    dx, dy = torch.from_numpy(x).to(dev), torch.empty((V, F), device=dev)
    other = torch.ones(48 << 18, device=dev)          # 48 MB
    sink = torch.zeros(1, device=dev)
    run_stream = torch.cuda.Stream()
    for name, (p, i) in (("locality reorder", (rptr, ridx)), ("no reorder", (ptr, idx)), ("uniform-random ids", (ptr, uid))):
        agg = gnc.Aggregator_GCN(torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev), torch.ones(E, device=dev), F, F)
        agg.schedule_balanced(0)
        out = {"input": name}
        run_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(run_stream):
            for label, between in (("back_to_back_us", False), ("with_48MB_read_between_us", True), ("back_to_back_again_us", False)):
                for _ in range(10):
                    agg.run(dx, dy, 512, "balanced")
                ts = []
                for _ in range(60):
                    if between:
                        sink += other.sum()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    agg.run(dx, dy, 512, "balanced")
                    b.record()
                    ts.append((a, b))
                torch.cuda.synchronize()
                per = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
                out[label] = round(per[len(per) // 2], 2)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "l2":
        l2_between_launches()
    elif len(sys.argv) > 1 and sys.argv[1] == "xcd":
        xcd_mapping()
    elif len(sys.argv) > 1 and sys.argv[1] == "host":
        host_or_device()
    else:
        main()
