#!/usr/bin/env python3
"""bench.py -- aggregated edges/s of the GCN SpMM hot path (feat = 128) on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself (fresh child processes through
torch.distributed.run, before anything touches the GPU) and relays their one line.  Under a launcher every rank process is a thin
supervisor that never touches the GPU either: it runs the real rank in a fresh child on the C-ABI RCCL step and, if that child fails
(oracle check of its first step, a watchdog on a hung exchange, a crash), runs it again in another fresh child on
torch.distributed's all_to_all_single -- the stated fallback; the line says which transport produced the number.

A step = one pass of the hot path (one aggregation Y = A.X) over the whole synthetic input, inputs
resident in HBM.  N = 1: BASELINE.json configs[1], arxiv-shaped CSR (169 343 x 1 166 243, seed 123),
GCN sum with explicit unit weights (Figure7/our.py:78), fp32, loaded with the locality reorder
applied like src/data.cu:96-133 (the un-reordered number is reported beside it).  N > 1: weak
scaling -- the global graph is N x arxiv-shaped, 1-D row-partitioned; every step pulls the halo
feature rows with one RCCL all-to-all and then runs the same kernel (gnn_computing_amd/dist.py).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

np = torch = None   # imported in main(), after the decision to launch / supervise ranks (those paths stay off the GPU stack)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FEAT = 128
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
L2_PEAK_GBPS = 34500.0  # same guide, "L2": 4 MiB per XCD, ~34.5 TB/s aggregate (the 2-D blocked order gathers from L2)
L2_GATHER_MEASURED_GBPS = 24500.0  # measured here: 256-byte row gathers from an XCD's own L2 (profiles/r02/gather_ceiling.txt)
PROFILE_ROUND = "r06"


def algorithmic_bytes(V, E, F, explicit_val=True):
    """SURVEY.md 8d gather model: E*(4F + 4 [idx] + 4 [val]) + V*4F [Y] + (V+1)*4 [ptr]."""
    return E * (4 * F + 4 + (4 if explicit_val else 0)) + V * 4 * F + (V + 1) * 4


def compulsory_bytes(V, E, F, explicit_val=True):
    """SURVEY.md 8d lower bound: every X row read once, every Y row written once, the CSR streamed once:
    2*V*4F + E*(4 [idx] + 4 [val]) + (V+1)*4."""
    return 2 * V * 4 * F + E * (4 + (4 if explicit_val else 0)) + (V + 1) * 4


def lib_md5():
    """md5 (first 12 hex digits) of the libgnnagg.so this process loads -- the label scripts/profile_round.sh stamps on its
    counter passes."""
    import hashlib
    p = os.environ.get("GNNAGG_LIB") or os.path.join(ROOT, "gnn_computing_amd", "libgnnagg.so")
    try:
        return hashlib.md5(open(p, "rb").read()).hexdigest()[:12]
    except OSError:
        return None


def pmc_traffic(tag):
    """Fabric-side bytes per launch of the config's kernels from the committed rocprofv3 PMC passes (scripts/profile_round.sh
    -> scripts/prof_config.sh -> profiles/<round>/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes, the
    gfx950 correction of MI355X_MICROARCH.md).  Returns (dominant kernel's bytes, all kernels' bytes per step, label, stale):
    `stale` is True when the build the counters were collected on is not the library running now."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        if os.path.exists(f):
            d = json.load(open(f)).get(tag)
            if d:
                label = d.get("_label", "")
                build = d.get("build") or (label.split("build ")[1].split(";")[0].strip() if "build " in label else None)
                # (the file's own `_label` says how the counters were collected and corrected; the line only names the file and the build)
                return (d.get("hbm_bytes_per_launch"), d.get("all_kernels_bytes_per_step"),
                        "profiles/%s/pmc_traffic.json [%s], build %s" % (rnd, tag, build), build != lib_md5())
    return None, None, None, None



def load_with_locality_reorder(name, ptr, idx):
    """The locality reorder APPLIED ON LOAD, the reference's way (load_graph(..., "_thres_0.2"), src/data.cu:96-133; our.py:79):
    gnn_computing_amd.graph.reorder_on_load writes the graph in the reference's cache format and the permutation of the library's
    generator as <dset>.reorder_thres_0.2, and the library's loader (gnnagg_load_graph) reads both and calls its reorderCSR.  The
    permutation file is kept per box (keyed by the library build): generating it takes tens of seconds on the products-shaped graph.
    Returns (ptr, idx, rows, seconds spent generating [0.0 on a cache hit], seconds spent loading)."""
    import gnn_computing_amd as gnc
    return gnc.graph.reorder_on_load(name, np.ascontiguousarray(ptr, np.int32), np.ascontiguousarray(idx, np.int32), key=lib_md5() or "x")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def time_steps(step_fn, steps, warmup, barrier):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides.
    Returns (wall seconds for the K steps, avg device seconds per step from ONE pair of HIP events around the K launches on
    the launch stream, median device seconds per step).  Nothing but the K steps is enqueued inside the timed region: an
    event record between launches is a barrier packet that drains the queue (it cost ~3 us per 80 us step); the per-step
    median comes from a separate pass with one event pair per launch, after the timed region."""
    # Host hygiene first (nothing of it is inside the timed region): the GPU boxes run this process in a CPU-quota'd container,
    # where a host thread that has been busy-waiting can be descheduled for 45-55 ms (seen once per few hundred launches,
    # scripts/history/exp_stall_hunt.py: GPU times unaffected); an idle period lets the quota window roll over, and the collector stays
    # off while the K launches are issued.  The W warm-up steps come AFTER it and run straight into the timed region: a device
    # left idle between the warm-up and the first timed launch -- even for the few milliseconds of a gc.collect() -- starts
    # the K steps cold (scripts/history/exp_timed_region.py, K = 20: 80.3 us per step with the idle period after the warm-up, 75.2 us
    # with the warm-up after it).
    # The launches run on ONE NON-NULL stream made for them; inputs, plans and oracle checks stay on the default stream (round 6,
    # tests/perf_reorder_discrepancy.py, tests/perf_side_stream_alloc.py).  On the null stream HIP orders every launch
    # against the process's other streams: once any exists (the rows mode forks two, a framework always has some) back-to-back null-stream
    # launches no longer overlap head to tail and a 74 us launch costs 79-87 us on the device; a dedicated stream keeps 73.8 us.  (Putting
    # EVERYTHING under a side stream -- allocations and uploads too -- measured 78.7 us: more active streams, the same loss.)
    if not getattr(time_steps, "_inside", False) and os.environ.get("BENCH_STREAM", "side") != "null":
        time_steps._inside = True
        try:
            s_ = getattr(time_steps, "_stream", None) or torch.cuda.Stream()
            time_steps._stream = s_
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                r_ = time_steps(step_fn, steps, warmup, barrier)
            torch.cuda.current_stream().wait_stream(s_)
            return r_
        finally:
            time_steps._inside = False
    import gc
    gc.collect()
    time.sleep(0.15)
    gc_was_on = gc.isenabled()
    gc.disable()
    for _ in range(warmup):
        step_fn()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step_fn()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    dev = ev0.elapsed_time(ev1) * 1e-3 / steps
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record()
    for i in range(steps):
        step_fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e-3 for i in range(steps))
    return wall, dev, per[len(per) // 2]


CPU_BASELINE_CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as orc
d = np.load(sys.argv[2])
ptr, idx, val, x = d["ptr"], d["idx"], d["val"], d["x"]
budget_s = float(sys.argv[3])
hw = orc.num_threads()
orc.gcn_seq(ptr, idx, val, x)  # warm-up (page-in, thread pool)
best_n, best_t = hw, float("inf")
for n in sorted({hw, max(hw // 2, 1), max(hw // 4, 1), min(hw, 32), min(hw, 16), min(hw, 8)}, reverse=True):
    orc.set_num_threads(n)
    orc.gcn_seq(ptr, idx, val, x)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        orc.gcn_seq(ptr, idx, val, x)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))  # the median, as the measurement below: a lucky pass must not pick an oversubscribed count
    if med < best_t:
        best_n, best_t = n, med
orc.set_num_threads(best_n)
times = []
t_end = time.perf_counter() + budget_s
while time.perf_counter() < t_end and len(times) < 300:
    t0 = time.perf_counter()
    orc.gcn_seq(ptr, idx, val, x)
    times.append(time.perf_counter() - t0)
print(json.dumps({"median_s": float(np.median(times)), "passes": len(times), "threads": best_n, "hardware_threads": hw}))
"""


def cpu_baseline(ptr, idx, val, x, budget_s=10.0):
    """The oracle (port of aggr_gcn.h:13-35; OpenMP over rows, `schedule(dynamic, 64)`; a row's 128 columns in eight AVX-512 accumulators --
    64 in AVX2 ones on a host without AVX-512 --, the next edges' rows prefetched) timed on the host cores:
    whole passes over the same arxiv-shaped workload, in a CHILD process with the threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores:
    SURVEY 8d) -- the pinning must not touch this process, whose main thread issues the timed GPU launches.  A quick sweep picks the
    thread count first (all hardware threads is rarely the fastest for a 90 MB gather working set), then the rest of the budget is
    measured."""
    import subprocess
    import tempfile
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        f = os.path.join(td, "w.npz")
        np.savez(f, ptr=np.ascontiguousarray(ptr, np.int32), idx=np.ascontiguousarray(idx, np.int32), val=np.ascontiguousarray(val, np.float32),
                 x=np.ascontiguousarray(x, np.float32))
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
        env.pop("OMP_NUM_THREADS", None)
        r = subprocess.run([sys.executable, "-c", CPU_BASELINE_CHILD, ROOT, f, str(budget_s)], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + r.stderr[-2000:])
    m = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": len(idx) / m["median_s"], "unit": "edges/s", "cores": m["threads"], "kind": "port", "cpu_model": model,
            "hardware_threads": m["hardware_threads"],
            "sample": "%d full passes of the same workload, median %.2f ms, %d of %d threads (best of a sweep), pinned; DESIGN.md 5" % (
                m["passes"], m["median_s"] * 1e3, m["threads"], m["hardware_threads"])}


def run_single(args, dev):
    import gnn_computing_amd as gnc
    mode = os.environ.get("BENCH_MODE", "balanced")
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    V, E = len(ptr) - 1, len(idx)
    rng = np.random.default_rng(123)
    x = rng.standard_normal((V, FEAT), dtype=np.float32)
    val = np.ones(E, np.float32)

    prep = {}

    def build(p, i):
        dp, di, dv = torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(val).to(dev)
        agg = gnc.Aggregator_GCN(dp, di, dv, FEAT, FEAT)
        torch.cuda.synchronize()
        t_s = time.perf_counter()  # schedule construction is reported separately, like the reference's
        if mode == "balanced":     # neighbor_grouping_schedule_time (graph_schedule.h:125-127)
            agg.schedule_balanced(0)
        elif mode == "scheduled":
            agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
            agg.schedule(gnc.Schedule.neighbor_grouping, [int(os.environ.get("BENCH_NG", "32"))])
        torch.cuda.synchronize()
        prep["schedule_prep_s"] = time.perf_counter() - t_s
        return agg

    dx = torch.from_numpy(x).to(dev)
    y = torch.empty((V, FEAT), dtype=torch.float32, device=dev)
    results = {}
    # without reorder
    agg0 = build(ptr, idx)
    results["no_reorder"] = time_steps(lambda: agg0.run(dx, y, 512, mode), args.steps, args.warmup, lambda: None)
    # with the locality reorder applied on load (reference: load_graph(..., "_thres_0.2"), our.py:79)
    # the reorder GENERATOR (what writes a <dset>.reorder_thres_0.2 file; the reference's is the offline script script/cluster2.py) is the
    # library's gnnagg_cluster_reorder_ex; the file is applied by the library's loader (load_with_locality_reorder above).  The
    # reference's MinHash-64 + LSH(0.2) + capped-64 clustering written in first-member order is timed beside it.
    nptr, nidx, rows, t_reorder, t_load = load_with_locality_reorder("arxiv", ptr, idx)
    agg1 = build(nptr, nidx)
    dx1 = torch.from_numpy(np.ascontiguousarray(x[rows])).to(dev)
    results["reorder"] = time_steps(lambda: agg1.run(dx1, y, 512, mode), args.steps, args.warmup, lambda: None)
    y_reorder = y.clone()
    rows_m, _ = gnc.cluster_reorder(ptr, idx, order="first_member")
    mptr, midx, _ = gnc.reorder_csr(ptr, idx, rows_m)
    agg2 = build(mptr, midx)
    dx2 = torch.from_numpy(np.ascontiguousarray(x[rows_m])).to(dev)
    results["reorder_minhash_clusters"] = time_steps(lambda: agg2.run(dx2, y, 512, mode), args.steps, args.warmup, lambda: None)
    y.copy_(y_reorder)
    # sanity: the timed kernel's output matches the oracle on this very input (outside the timed region)
    from oracle import oracle as orc
    ps, ix, tg = agg1.get_schedule(mode) if mode != "rows" else (None, None, None)
    seg = agg1.mode_params(mode)[1] if mode != "rows" else 0
    ref = (orc.gcn_seq(nptr, nidx, val, x[rows]) if mode == "rows"
           else orc.gcn_grouped(ps, tg, nidx, val, x[rows], V, seg=seg))
    assert np.array_equal(y.cpu().numpy(), ref), "bench output differs from the oracle"

    which = os.environ.get("BENCH_HEADLINE", "reorder")
    wall, dev_s, med_s = results[which]
    ms = wall / args.steps * 1e3
    B = algorithmic_bytes(V, E, FEAT)
    C = compulsory_bytes(V, E, FEAT)
    achieved = B / dev_s / 1e9
    # The roofline this launch can be held to.  X (86.7 MB) is Infinity-Cache resident, so gather bytes / time is cache + HBM
    # delivery and may exceed the 8 TB/s HBM figure: the ceiling is MEASURED instead, in this process, under the same timing
    # protocol -- the probe launch (gnnagg_gcn_probe_gather) issues this kernel's descriptor / id / value loads and its
    # 512-byte row gathers (same addresses, same batching) with no FMA chain and no store.  frac = probe time / kernel time.
    agg_h, x_h = (agg1, dx1) if which == "reorder" else (agg0, dx)
    _, probe_s, probe_med = time_steps(lambda: agg_h.probe_gather(x_h, mode), args.steps, args.warmup, lambda: None)
    # a second ceiling that owes nothing to the graph: same degrees, neighbor ids drawn uniformly (no locality at all)
    rid = torch.from_numpy(np.random.default_rng(7).integers(0, V, E).astype(np.int32)).to(dev)
    agg_u = gnc.Aggregator_GCN(torch.from_numpy(ptr).to(dev), rid, torch.from_numpy(val).to(dev), FEAT, FEAT)
    if mode == "balanced":
        agg_u.schedule_balanced(0)
    elif mode == "scheduled":
        agg_u.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg_u.schedule(gnc.Schedule.neighbor_grouping, [int(os.environ.get("BENCH_NG", "32"))])
    _, probe_u_s, _ = time_steps(lambda: agg_u.probe_gather(dx, mode), args.steps, args.warmup, lambda: None)
    _, kern_u_s, _ = time_steps(lambda: agg_u.run(dx, y, 512, mode), args.steps, args.warmup, lambda: None)
    traffic, _, traffic_label, traffic_stale = pmc_traffic("A")
    # ceilings that bound the bytes `achieved` divides: 512-byte row gathers, ids uniform over a window the size
    # of X (Infinity-Cache resident, like this input's X) and over 5 GB (HBM), measured in this process
    ceil_mall, ceil_hbm = gather_ceiling(dev, "g512_mall"), gather_ceiling(dev, "g512_hbm")
    other = "no_reorder" if which == "reorder" else "reorder"
    # roofline.  bound = HBM / fabric bandwidth, 8 TB/s.  `achieved` = ALGORITHMIC (gather-model) bytes over the kernel's
    # average launch time -- on this input most of those bytes are served by L2 / Infinity Cache, so it can exceed the peak
    # (`algorithmic_frac` > 1) and is NOT a utilisation.  `frac` = what the counters say crossed the fabric (`traffic`, committed
    # profile) over the same time over 8 TB/s; `frac_vs_gather_ceiling` divides the gather-model rate by the rate measured in this process
    # for 512-byte row gathers uniform over an X-sized window; `probe_frac` = this launch's own gather pattern without chains or stores.
    # What every field is, in full sentences: DESIGN.md section 5 ("fields of the bench line") -- the line itself stays short, the
    # driver's record truncates strings.
    traffic_gbps = traffic / dev_s / 1e9 if traffic else None
    out = {
        "metric": "aggregated edges/sec, GCN SpMM feat=128", "value": E / (wall / args.steps), "unit": "edges/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "arxiv-shaped CSR 169343x1166243 (seed 123), GCN sum, feat=128, unit weights, %s, mode=%s" % (
                       "locality reorder applied on load (.reorder_thres_0.2)" if which == "reorder" else "no reorder", mode),
                   "num_v": V, "num_e": E, "feat": FEAT},
        "achieved_gbps": achieved,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (traffic_gbps / HBM_PEAK_GBPS) if traffic_gbps else achieved / HBM_PEAK_GBPS,
                     "frac_is": ("counter traffic (fabric side: Infinity-Cache hits included) / avg launch time / 8 TB/s; DESIGN.md 5"
                                 if traffic_gbps else "algorithmic bytes / avg launch time / 8 TB/s (no counter file); DESIGN.md 5"),
                     "frac_vs_gather_ceiling": achieved / ceil_mall["gbps"],
                     "frac_vs_gather_ceiling_is": "gather-model rate / measured uniform 512-B gather rate (X-sized window); locality may exceed 1",
                     "gather_ceiling": {"gbps": ceil_mall["gbps"], "us": ceil_mall["us"]},
                     "gather_ceiling_hbm": {"gbps": ceil_hbm["gbps"], "us": ceil_hbm["us"]},
                     "traffic_frac_of_mall_gather_ceiling": (traffic_gbps / ceil_mall["gbps"]) if traffic_gbps else None,
                     "traffic": traffic, "traffic_source": traffic_label, "traffic_stale": traffic_stale,
                     "traffic_gbps": traffic_gbps,
                     "algorithmic_frac": achieved / HBM_PEAK_GBPS,
                     "probe_frac": probe_s / dev_s, "probe_gbps": B / probe_s / 1e9,
                     "kernel": "k_gcn_plan", "algorithmic_bytes": B, "compulsory_bytes": C,
                     "avg_launch_us": dev_s * 1e6, "median_launch_us": med_s * 1e6,
                     "ceiling_probe_us": probe_s * 1e6, "ceiling_probe_median_us": probe_med * 1e6,
                     "uniform_random_ids": {"kernel_us": kern_u_s * 1e6, "probe_us": probe_u_s * 1e6},
                     "gather_gbps": achieved, "hbm_peak_gbps": HBM_PEAK_GBPS,
                     "gather_frac_of_hbm_peak": achieved / HBM_PEAK_GBPS,
                     "compulsory_gbps": C / dev_s / 1e9, "compulsory_frac_of_hbm_peak": C / dev_s / 1e9 / HBM_PEAK_GBPS},
        other: {"value": E / (results[other][0] / args.steps), "avg_launch_us": results[other][1] * 1e6,
                "achieved_gbps": B / results[other][1] / 1e9},
        "reorder_minhash_clusters": {"value": E / (results["reorder_minhash_clusters"][0] / args.steps),
                                     "avg_launch_us": results["reorder_minhash_clusters"][1] * 1e6},
        "reorder_prep_s": t_reorder, "reorder_load_s": t_load, "reorder_cache_hit": t_reorder == 0.0,
        "reorder_walkers": int(os.environ.get("GNNAGG_REORDER_WALKERS", "0")) or (64 if E >= 20000000 else 1),
        "schedule_prep_s": prep.get("schedule_prep_s"),
    }
    if mode == "balanced" and which == "reorder":
        del agg0, agg2, agg_u, dx, dx2, rid
        torch.cuda.empty_cache()
        out["configs"] = other_configs(args, dev, nptr, nidx, val, x[rows])
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(nptr, nidx, val, x[rows], args.cpu_budget)
    out["summary"] = summarize(out)   # LAST key of the line, <= 1 KB: what the driver's 2 000-character tail must show
    return out


def summarize(out):
    """Every record of the line in one compact object: name -> [ms_per_step, roofline.frac, frac_vs_gather_ceiling, verified_rows]
    (A: the headline, whole output compared; *_no_reorder: the arm without the locality reorder)."""
    def row(rec, verified_rows):
        r = rec.get("roofline", {})
        return [round(rec["ms_per_step"], 5), round(r.get("frac", 0.0), 4), round(r.get("frac_vs_gather_ceiling", 0.0), 4), verified_rows]
    s = {"fields": ["ms_per_step", "roofline.frac", "frac_vs_gather_ceiling", "verified_rows"],
         "A": row(out, out["config"]["num_v"])}
    if "no_reorder" in out:
        s["A_no_reorder_us"] = round(out["no_reorder"]["avg_launch_us"], 2)
    ok = True
    for name, c in (out.get("configs") or {}).items():
        if "error" in c:
            s[name], ok = "error", False
            continue
        s[name] = row(c, c.get("verified_rows", 0))
        ok = ok and c.get("verified_against_oracle") is True
        nr = c.get("no_reorder")
        if nr:
            s[name + "_no_reorder"] = [round(nr["ms_per_step"], 5), round(nr.get("frac", 0.0), 4), round(nr.get("frac_vs_gather_ceiling", 0.0), 4),
                                       nr.get("verified_rows", 0)]
            ok = ok and nr.get("verified_against_oracle") is True
    s["edges_per_s"] = round(out["value"])
    if "cpu_baseline" in out:
        s["cpu_edges_per_s"] = round(out["cpu_baseline"]["value"])
    s["verified"] = ok
    return s


def pick_rows(ptr_h, k, seed, hubs=3):
    """k random rows + the `hubs` heaviest rows + an empty row (what tests/test_gpu_fullsize.py samples)."""
    deg = np.diff(ptr_h)
    rng = np.random.default_rng(seed)
    rows = set(rng.integers(0, len(deg), k).tolist())
    rows.update(np.argsort(deg)[-hubs:].tolist())
    empties = np.nonzero(deg == 0)[0]
    if len(empties):
        rows.add(int(empties[0]))
    return np.array(sorted(rows), np.int64)


def sample_rows(ptr_h, idx, rows):
    """Sub-CSR of the chosen rows (numpy), column ids unchanged; also the edge positions of those rows."""
    sub_ptr = np.zeros(len(rows) + 1, np.int32)
    sub_ptr[1:] = np.cumsum(ptr_h[rows + 1] - ptr_h[rows])
    parts = [idx[int(ptr_h[r]):int(ptr_h[r + 1])] for r in rows]
    sub_idx = torch.cat(parts).cpu().numpy() if parts else np.empty(0, np.int32)
    eids = np.concatenate([np.arange(ptr_h[r], ptr_h[r + 1]) for r in rows]).astype(np.int64)
    return sub_ptr, sub_idx, eids


def sum_rows_f64(sp, si, xh, vh=None, w=None, heads=1):
    """Float64 value of the sampled rows' aggregation (the truth the fp32 orders are judged against): sum_e v_e x[idx_e], or with
    per-edge per-head weights w[E', H].  Chunked so that a 800 k-edge hub row never materialises more than 64 k gathered rows."""
    n, F = len(sp) - 1, xh.shape[1]
    out = np.zeros((n, F))
    for k in range(n):
        for e0 in range(int(sp[k]), int(sp[k + 1]), 65536):
            e1 = min(e0 + 65536, int(sp[k + 1]))
            g = xh[si[e0:e1]].astype(np.float64)
            if w is not None:
                g = g.reshape(e1 - e0, heads, F // heads) * w[e0:e1, :, None]
                out[k] += g.sum(axis=0).reshape(F)
            elif vh is not None:
                out[k] += (g * vh[e0:e1, None].astype(np.float64)).sum(axis=0)
            else:
                out[k] += g.sum(axis=0)
    return out


def verify_config(cfg, agg, ptr, idx, x, y, att=None, val=None, heads=1):
    """The timed step's output against the oracle on a sample of rows (random rows + the heaviest hubs + an empty row): per-row
    results depend only on that row's edges, so a row sample is exact where the whole 115 M-edge pass would take the oracle minutes.
    R / P1: BIT-EQUAL to the library's order restated by the oracle (orc.locality_schedule / neighbor_grouping + gcn_grouped), and inside
    north_star's 1e-5 * sum_e |v_e x_e| of the float64 value.  G: inside 1e-5 (condition-aware) of the restated order (orc.gat_grouped)
    and of the float64 edge-softmax.  The reference's own CSR-order fp32 chain (orc.gcn_seq / gat_fused) is measured against the same
    float64 value and reported as `reference_order_worst_ratio`: on an 800 k-edge hub row the sequential chain itself sits at 0.7 - 1.1
    of that bound, so it cannot serve as the yardstick there.  Raises on a mismatch."""
    from oracle import oracle as orc
    ptr_h = ptr.cpu().numpy()
    rows = pick_rows(ptr_h, 40 if cfg != "P1" else 200, 1)
    sp, si, eids = sample_rows(ptr_h, idx, rows)
    xh = x.cpu().numpy()
    got = y[torch.from_numpy(rows).to(y.device)].cpu().numpy()
    chunk, seg = agg.balanced_params()
    parts = agg.balanced_partitions()
    n = len(rows)
    if cfg in ("R", "P1"):
        if cfg == "R":
            vh = None
            ps, ix, tg, _ = orc.locality_schedule(sp, si, parts, agg.balanced_partition_columns(), ng=chunk)
            div = np.maximum(np.diff(sp), 1)[:, None].astype(np.float32)
            restated, chain = orc.gcn_grouped(ps, tg, ix, None, xh, n, seg=0) / div, orc.gcn_mean(sp, si, None, xh)
            how = "bit-equal to the restated 2-D blocked order"
        else:
            vh = val[torch.from_numpy(eids).to(val.device)].cpu().numpy() if val is not None else None
            ps, tg = orc.neighbor_grouping(sp, chunk)
            div = np.ones((n, 1), np.float32)
            restated, chain = orc.gcn_grouped(ps, tg, si, vh, xh, n, seg=seg), orc.gcn_seq(sp, si, vh, xh)
            how = "bit-equal to the restated chunked order"
        exact = np.array_equal(got, restated)
        truth = sum_rows_f64(sp, si, xh, vh) / div
        bound = 1e-5 * orc.gcn_abs_scale(sp, si, vh, xh).astype(np.float64) / div + 1e-30
        ratio, ref_ratio = float((np.abs(got - truth) / bound).max()), float((np.abs(chain - truth) / bound).max())
        ok = exact and ratio <= 1.0
        how += "; within 1e-5 * sum|v x| of float64; DESIGN.md 5"
    else:
        H, F = heads, x.shape[1]
        atth = att.cpu().numpy()
        att_mix = atth.copy()                      # centre term of compact row k at [k,:,0], source terms at [id,:,1]
        att_mix[:n, :, 0] = atth[rows, :, 0]
        ps, ix, tg, _ = orc.locality_schedule(sp, si, parts, agg.balanced_partition_columns(), ng=chunk)
        restated, _, _ = orc.gat_grouped(ps, tg, ix, att_mix, xh, n, H, seg=0)
        # float64 edge softmax of the sampled rows (aggr_gat.h:125-163 in exact arithmetic, slope 0.2)
        sc = np.repeat(atth[rows, :, 0].astype(np.float64), np.diff(sp), axis=0) + atth[si, :, 1].astype(np.float64)
        w = np.exp(np.maximum(sc, 0.2 * sc))
        den = np.add.reduceat(w, sp[:-1][np.diff(sp) > 0], axis=0)
        nz = np.diff(sp) > 0
        wn = w / np.repeat(den, np.diff(sp)[nz], axis=0)
        truth = sum_rows_f64(sp, si, xh, w=wn, heads=H)
        scale = sum_rows_f64(sp, si, np.abs(xh), w=wn, heads=H)
        bound = 1e-5 * (scale + np.abs(truth)) + 1e-30
        ratio = float(max((np.abs(got - truth) / bound).max(), (np.abs(got.astype(np.float64) - restated) / bound).max()))
        ref_ratio = float((np.abs(orc.gat_fused(sp, si, att_mix, xh, H) - truth) / bound).max())
        ok = ratio <= 1.0 and not np.isnan(got).any()
        how = "within 1e-5 * (sum w|x| + |y|) of the restated blocked order and of the float64 edge softmax; DESIGN.md 5"
    if not ok:
        raise RuntimeError("config %s: the timed step's output differs from the oracle on the sampled rows (worst ratio to the bound %.3g)" % (cfg, ratio))
    return {"verified_against_oracle": True, "verified_rows": int(n), "verified_how": how, "worst_ratio_to_1e-5_bound": ratio,
            "reference_order_worst_ratio": ref_ratio}


_CEILINGS = {}


def gather_ceiling(dev, key):
    """Measured in THIS process (gnnagg_probe_row_gather, a few ms each): what the memory system offers to row gathers in the kernels'
    own access shape when the gathered rows live in one level -- the denominators of `frac_vs_gather_ceiling`."""
    if key not in _CEILINGS:
        import gnn_computing_amd as gnc
        seg, pitch, window, private = {
            "g512_mall": (512, 512, 169343 * 512, False),    # config A: X = 86.7 MB, Infinity-Cache resident
            "g512_hbm": (512, 512, 5 << 30, False),          # 512-B rows from a 5 GB window: HBM
            "g256_l2": (256, 256, 2 << 20, True),            # R / G: 256-B tile rows out of each XCD's own L2 (2 MB slice each)
            "g400_hbm": (400, 400, 5 << 30, False),          # P1: 400-B rows at a 400-B pitch, X = 980 MB: HBM
        }[key]
        _CEILINGS[key] = gnc.probe.row_gather_ceiling(dev, seg, pitch, window, private)
        torch.cuda.empty_cache()
    return _CEILINGS[key]


def measure_config(cfg, args, dev, graph=None):
    """One of the other single-GPU configurations of BASELINE.json, timed like the headline (time_steps) and checked against the oracle
    on sampled rows: R = reddit-shaped SAGE mean F=602, G = reddit-shaped GAT 8x32, P1 = products-shaped GCN F=100 on one GPU.
    Reference call sequences being timed: Figure9/main.cu:59-74 (GCN run), Figure10/main_a.cu:82-110 (GAT run)."""
    import gnn_computing_amd as gnc
    name = {"R": "reddit", "G": "reddit", "P1": "products"}[cfg]
    ptr, idx = graph if graph is not None else gnc.graph.dataset(name, device=dev)
    V, E = ptr.numel() - 1, idx.numel()
    H, att, val = 1, None, None
    if cfg == "R":
        F, what = 602, "reddit-shaped CSR %dx%d, GraphSAGE mean, feat=602, mode=balanced" % (V, E)
        agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
        x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
        B = E * (4 * F + 4) + V * 4 * F + 4 * (V + 1)
        kernel, ceil_key = "k_gcn_span (+ k_tile_x, k_combine_groups)", "g256_l2"
    elif cfg == "G":
        H, F = 8, 256
        what = "reddit-shaped CSR %dx%d, GAT 8 heads x 32 fused, mode=balanced" % (V, E)
        agg = gnc.Aggregator_GAT(ptr, idx, F, F)
        x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
        att = torch.randn((V, H, 2), device=dev)
        B = E * (4 * F + 4 + 4 * H) + V * (4 * F + 4 * H) + 4 * (V + 1)
        kernel, ceil_key = "k_gat_span (+ k_tile_x, k_combine_groups_gat)", "g256_l2"
    else:
        F, what = 100, "products-shaped CSR %dx%d, GCN sum, feat=100, unit weights, mode=balanced, 1 GPU" % (V, E)
        val = torch.ones(E, device=dev)
        agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
        x, y = torch.randn((V, F), device=dev), torch.empty((V, F), device=dev)
        B = algorithmic_bytes(V, E, F)
        kernel, ceil_key = "k_gcn_plan", "g400_hbm"
    steps, warm = min(args.steps, 20), min(args.warmup, 3)
    explicit = cfg == "P1"
    C = (2 * V * 4 * F + E * (4 + (4 if explicit else 0)) + (V + 1) * 4) + (V * 8 * H * 2 if cfg == "G" else 0)

    def measure_arm(agg, ptr_a, idx_a, tag):
        """time_steps + oracle check + gather probe + roofline of one aggregator over one numbering of the graph"""
        if cfg == "G":
            step = lambda: agg.run(x, att, y, 128, "balanced", heads=H)  # noqa: E731
            probe = lambda: agg.probe_gather(x, att, "balanced", heads=H)  # noqa: E731
        else:
            step = lambda: agg.run(x, y, 512, "balanced", reduce="mean" if cfg == "R" else "sum")  # noqa: E731
            probe = lambda: agg.probe_gather(x, "balanced")  # noqa: E731
        # the first call builds the library-chosen order (on the device: plan_gpu.hip) and reserves the scratch; reported separately, like
        # the reference's neighbor_grouping_schedule_time (graph_schedule.h:125-127)
        torch.cuda.synchronize()
        t_first = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t_first
        plan = agg.plan_info()
        wall, dev_s, med_s = time_steps(step, steps, warm, lambda: None)
        verified = verify_config(cfg, agg, ptr_a, idx_a, x, y, att=att, val=val, heads=H)
        achieved = B / dev_s / 1e9
        _, probe_s, _ = time_steps(probe, steps, warm, lambda: None)
        traffic, traffic_step, traffic_label, traffic_stale = pmc_traffic(tag)
        blocked = agg.balanced_partitions() > 1
        ceil = gather_ceiling(dev, ceil_key if blocked or cfg == "P1" else "g512_hbm")
        # roofline.  P1 (chunked plan, X far larger than the caches): HBM / fabric bound, frac = counter traffic of the step / step time /
        # 8 TB/s.  R and G (2-D blocked order: the gathered tile rows are served by the XCDs' L2s): the bound is the L2, frac = gather-model
        # bytes / step time / 34.5 TB/s (the guide's L2 figure).  `frac_vs_gather_ceiling` divides the same gather-model rate by the rate
        # MEASURED in this process for the same segment size with ids UNIFORM over a window in the level the rows live in -- an input with
        # locality (hot sources, a reordered graph) can exceed it; `probe_frac` is the bounding ratio of the launch itself.
        if blocked:
            bound, peak, frac = "l2", L2_PEAK_GBPS, achieved / L2_PEAK_GBPS
            frac_is = "gather-model bytes / step time / 34.5 TB/s (guide's aggregate L2 figure); DESIGN.md 5"
        else:
            bound, peak = "hbm", HBM_PEAK_GBPS
            frac = (traffic_step / dev_s / 1e9 / HBM_PEAK_GBPS) if traffic_step else achieved / HBM_PEAK_GBPS
            frac_is = ("counter traffic of the step (fabric side: Infinity-Cache hits included) / step time / 8 TB/s; DESIGN.md 5" if traffic_step
                       else "algorithmic bytes / step time / 8 TB/s (no counter file); DESIGN.md 5")
        rec = {"value": E / (wall / steps), "ms_per_step": wall / steps * 1e3, "achieved_gbps": achieved,
               "schedule_prep_s": plan["plan_s"], "first_call_s": t_first, "plan_bytes": plan["plan_bytes"], "scratch_bytes": plan["scratch_bytes"],
               "source_partitions": agg.balanced_partitions(),
               "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": frac, "frac_is": frac_is,
                            "frac_vs_gather_ceiling": achieved / ceil["gbps"],
                            "gather_ceiling": {"gbps": ceil["gbps"], "us": ceil["us"]},
                            "frac_vs_gather_ceiling_is": "gather-model rate / measured UNIFORM-id gather rate of the same segment size; locality may exceed 1",
                            "traffic": traffic, "traffic_step": traffic_step, "traffic_source": traffic_label,
                            "traffic_stale": traffic_stale, "kernel": kernel, "algorithmic_bytes": B,
                            "compulsory_bytes": C, "avg_launch_us": dev_s * 1e6, "median_launch_us": med_s * 1e6,
                            "probe_frac": probe_s / dev_s, "ceiling_probe_us": probe_s * 1e6,
                            "hbm_peak_gbps": HBM_PEAK_GBPS, "gather_frac_of_hbm_peak": achieved / HBM_PEAK_GBPS,
                            "traffic_frac_of_hbm_peak": (traffic_step / dev_s / 1e9 / HBM_PEAK_GBPS) if traffic_step else None,
                            "compulsory_frac_of_hbm_peak": C / dev_s / 1e9 / HBM_PEAK_GBPS}}
        rec.update(verified)
        return rec

    arm = measure_arm(agg, ptr, idx, cfg)
    no_reorder = None
    if cfg == "P1" and os.environ.get("BENCH_P1_REORDER", "1") != "0":
        # north_star applies the locality reorder ON LOAD and SURVEY 8(e) partitions after it, so the line's `value` is the reordered graph
        # (like the headline); the generator's numbering rides beside it as `no_reorder`.  Both arms are checked
        # against the oracle on sampled rows of their own numbering.
        no_reorder = arm
        del agg
        torch.cuda.empty_cache()
        nptr, nidx, _, t_gen, t_load = load_with_locality_reorder("products", ptr.cpu().numpy(), idx.cpu().numpy())
        ptr_r, idx_r = torch.from_numpy(nptr).to(dev), torch.from_numpy(nidx).to(dev)
        agg = gnc.Aggregator_GCN(ptr_r, idx_r, val, F, F)
        arm = measure_arm(agg, ptr_r, idx_r, "P1_reorder")
        arm["with_locality_reorder"] = {"value": arm["value"], "ms_per_step": arm["ms_per_step"], "reorder_prep_s": t_gen, "reorder_load_s": t_load,
                                        "reorder_cache_hit": t_gen == 0.0, "verified_against_oracle": arm["verified_against_oracle"],
                                        "reorder_walkers": int(os.environ.get("GNNAGG_REORDER_WALKERS", "0")) or (64 if E >= 20000000 else 1)}
        what += ", locality reorder applied on load (.reorder_thres_0.2)"
        no_reorder = {k: no_reorder[k] for k in ("value", "ms_per_step", "verified_against_oracle", "verified_rows", "worst_ratio_to_1e-5_bound")} | {
            "frac": no_reorder["roofline"]["frac"], "frac_vs_gather_ceiling": no_reorder["roofline"]["frac_vs_gather_ceiling"],
            "avg_launch_us": no_reorder["roofline"]["avg_launch_us"], "probe_frac": no_reorder["roofline"]["probe_frac"],
            "traffic_step": no_reorder["roofline"]["traffic_step"]}
    rec = {"metric": "aggregated edges/sec, config %s" % cfg, "value": arm["value"], "unit": "edges/s",
           "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": arm["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": what, "num_v": V, "num_e": E, "feat": F, "source_partitions": arm["source_partitions"]}}
    rec.update({k: v for k, v in arm.items() if k not in ("value", "ms_per_step", "source_partitions")})
    rec.setdefault("with_locality_reorder", None)
    rec["no_reorder"] = no_reorder
    return rec


def run_other_config(args, dev):
    rec = measure_config(args.config, args, dev)
    r = rec["roofline"]
    rec["summary"] = {"fields": ["ms_per_step", "roofline.frac", "frac_vs_gather_ceiling", "verified_rows"],
                      args.config: [round(rec["ms_per_step"], 5), round(r["frac"], 4), round(r["frac_vs_gather_ceiling"], 4), rec["verified_rows"]],
                      "edges_per_s": round(rec["value"]), "verified": rec["verified_against_oracle"] is True}
    if rec.get("no_reorder"):
        nr = rec["no_reorder"]
        rec["summary"][args.config + "_no_reorder"] = [round(nr["ms_per_step"], 5), round(nr["frac"], 4), round(nr["frac_vs_gather_ceiling"], 4), nr["verified_rows"]]
    return rec


def measure_rows_mode(args, dev, nptr, nidx, val, x_rows, ceil):
    """The headline input in GNNAGG_MODE_ROWS: the literal aggr_gcn order (aggr_gcn.h:5-36), every row one sequential FMA chain in CSR
    order, bit-equal to the oracle's gcn_seq over the WHOLE output."""
    import gnn_computing_amd as gnc
    from oracle import oracle as orc
    V, E = len(nptr) - 1, len(nidx)
    agg = gnc.Aggregator_GCN(torch.from_numpy(nptr).to(dev), torch.from_numpy(nidx).to(dev), torch.from_numpy(val).to(dev), FEAT, FEAT)
    dx = torch.from_numpy(np.ascontiguousarray(x_rows)).to(dev)
    y = torch.empty((V, FEAT), dtype=torch.float32, device=dev)
    step = lambda: agg.run(dx, y, 512, "rows")  # noqa: E731
    step()
    torch.cuda.synchronize()
    plan = agg.plan_info()
    wall, dev_s, med_s = time_steps(step, args.steps, args.warmup, lambda: None)
    if not np.array_equal(y.cpu().numpy(), orc.gcn_seq(nptr, nidx, val, x_rows)):
        raise RuntimeError("rows mode differs from the oracle's CSR-order chains")
    B = algorithmic_bytes(V, E, FEAT)
    achieved = B / dev_s / 1e9
    traffic, traffic_step, traffic_label, traffic_stale = pmc_traffic("A_rows")
    return {"metric": "aggregated edges/sec, GCN SpMM feat=128, canonical rows mode", "value": E / (wall / args.steps), "unit": "edges/s",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "config": {"workload": "the headline input in GNNAGG_MODE_ROWS (aggr_gcn's own order: one CSR-order FMA chain per row)",
                       "num_v": V, "num_e": E, "feat": FEAT},
            "achieved_gbps": achieved, "schedule_prep_s": plan["plan_s"], "plan_bytes": plan["plan_bytes"], "scratch_bytes": plan["scratch_bytes"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (traffic_step / dev_s / 1e9 / HBM_PEAK_GBPS) if traffic_step else achieved / HBM_PEAK_GBPS,
                         "frac_is": ("counter traffic of the step (fabric side: Infinity-Cache hits included) / step time / 8 TB/s; DESIGN.md 5"
                                     if traffic_step else "algorithmic bytes / step time / 8 TB/s (no counter file); DESIGN.md 5"),
                         "frac_vs_gather_ceiling": achieved / ceil["gbps"],
                         "gather_ceiling": {"gbps": ceil["gbps"], "us": ceil["us"]},
                         "traffic": traffic, "traffic_step": traffic_step, "traffic_source": traffic_label, "traffic_stale": traffic_stale,
                         "kernel": "k_gcn_plan (short rows) beside k_gcn_rows_long (medium, hub rows) on forked streams",
                         "algorithmic_bytes": B, "compulsory_bytes": compulsory_bytes(V, E, FEAT),
                         "avg_launch_us": dev_s * 1e6, "median_launch_us": med_s * 1e6,
                         "probe_frac": None},
            "verified_against_oracle": True, "verified_rows": V,
            "verified_how": "whole output array_equal to orc.gcn_seq (aggr_gcn's chain order)"}


def other_configs(args, dev, nptr, nidx, val, x_rows):
    """The sub-records of the N = 1 line: every other single-GPU configuration BASELINE.json names, so that one driver command times
    them all.  A failure in one of them is recorded in its place and never takes the headline line down."""
    import gnn_computing_amd as gnc
    which = [c for c in os.environ.get("BENCH_CONFIGS", "A_rows,R,G,P1").split(",") if c]
    out, reddit = {}, None
    for cfg in which:
        t0 = time.perf_counter()
        try:
            if cfg == "A_rows":
                out[cfg] = measure_rows_mode(args, dev, nptr, nidx, val, x_rows, gather_ceiling(dev, "g512_mall"))
            elif cfg in ("R", "G"):
                if reddit is None:
                    reddit = gnc.graph.dataset("reddit", device=dev)
                out[cfg] = measure_config(cfg, args, dev, graph=reddit)
            elif cfg == "P1":
                reddit = None
                torch.cuda.empty_cache()
                out[cfg] = measure_config(cfg, args, dev)
            else:
                raise ValueError("unknown config %r" % cfg)
            out[cfg]["bench_wall_s"] = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            log("bench.py: sub-record %s failed: %r" % (cfg, e))
            out[cfg] = {"error": repr(e), "bench_wall_s": time.perf_counter() - t0}
        torch.cuda.empty_cache()
    return out


class Watchdog:
    """N > 1 only: a hung exchange (an RCCL kernel waiting for a peer that never posts) cannot be cancelled from inside the process.
    arm(seconds, what) starts a deadline; when it passes the process exits with code 18 and the supervising parent starts the
    fallback transport in fresh processes.  disarm() when the guarded phase is over."""

    def __init__(self):
        import threading
        self.deadline, self.what = None, ""
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        while True:
            time.sleep(0.5)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                log("bench.py watchdog: '%s' did not finish in time; exiting 18 (the supervisor falls back to the other transport)" % self.what)
                os._exit(18)

    def arm(self, seconds, what):
        self.what, self.deadline = what, time.monotonic() + seconds

    def disarm(self):
        self.deadline = None


def partitioned_run(args, dev, rank, world, strong, steps, warmup, transport, dog):
    """One row-partitioned configuration over the `world` ranks: weak (N x arxiv-shaped, feat 128) or strong (ONE products-shaped
    graph, feat 100).  Returns the measurements on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    import gnn_computing_amd as gnc
    from gnn_computing_amd.dist import PartitionedGCN
    feat = 100 if strong else FEAT
    V1, E1 = gnc.graph.SHAPES["products" if strong else "arxiv"]
    Vg, Eg = (V1, E1) if strong else (V1 * world, E1 * world)
    # rank 0 generates the global graph on its GPU (seeded, community order = "locality reorder applied on load"); every
    # rank receives the row offsets (to cut the same nnz-balanced partition) and ONLY ITS OWN rows' neighbor ids
    dog.arm(300, "graph generation + distribution of the row slices")
    if rank == 0:
        ptr_t, idx_t = gnc.graph.powerlaw_csr(Vg, Eg, seed=123, device=dev, community_order=True, p_local=1.0 - args.global_share)
    else:
        ptr_t = torch.empty(Vg + 1, dtype=torch.int32, device=dev)
    dist.broadcast(ptr_t, src=0)
    ptr = ptr_t.cpu().numpy()
    bounds = gnc.partition_rows(ptr, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    e0, e1 = int(ptr[r0]), int(ptr[r1])
    p2p_dev = dev if dist.get_backend() == "nccl" else torch.device("cpu")   # (gloo moves point-to-point data through the host)
    if rank == 0:
        for r in range(1, world):
            dist.send(idx_t[int(ptr[bounds[r]]):int(ptr[bounds[r + 1]])].contiguous().to(p2p_dev), dst=r)
        idx_slice = idx_t[e0:e1].cpu().numpy()
        del idx_t
    else:
        buf = torch.empty(e1 - e0, dtype=torch.int32, device=p2p_dev)
        dist.recv(buf, src=0)
        idx_slice = buf.cpu().numpy()
        del buf
    del ptr_t
    val_slice = np.ones(e1 - e0, np.float32)
    # Staged (pipelined) exchange by default: "auto" = stripes, one per 64 MB of halo rows, at most 4 (dist.py) -- the halo-source pass of stage s
    # runs under stage s + 1.  Default since round 6: the staged step now runs against an ASYNCHRONOUS peer (tests/fake_rccl: stream-ordered
    # copies, no host synchronisation inside a group) with the halo poisoned and a rank late on the device --
    # tests/test_gpu_dist.py::test_cabi_step_is_ordered_by_its_events_not_by_luck (stripe 2 / 3, owner),
    # ::test_the_late_peer_test_can_fail (its negative control) and ::test_cabi_step_replays_from_a_captured_graph_at_world_4.
    # BENCH_STAGES=1 | owner | K overrides.
    stages = os.environ.get("BENCH_STAGES", "auto")
    stages = int(stages) if stages.lstrip("-").isdigit() else stages
    t_plan = time.perf_counter()
    dog.arm(240, "communicator + plan exchange (%s transport)" % transport)
    pg = PartitionedGCN(ptr[r0:r1 + 1], idx_slice, val_slice, feat, device=dev, mode=os.environ.get("BENCH_MODE", "balanced"),
                        row_slice=True, bounds=bounds, num_cols=Vg, transport=transport, stages=stages)
    t_plan = time.perf_counter() - t_plan
    hx = pg.hx
    rccl_ranks = None
    if hx.rccl is not None:   # what the C-ABI's communicator itself says (gnnagg_dist_comm_info)
        import ctypes
        rr, ww = ctypes.c_int(-1), ctypes.c_int(-1)
        gnc._lib.check(gnc.lib().gnnagg_dist_comm_info(hx.rccl._h, ctypes.byref(rr), ctypes.byref(ww)))
        assert rr.value == rank
        rccl_ranks = ww.value
    # what carried the halo rows, stated by the library itself (gnnagg_dist_transport_info) and by the devices the ranks sit on: an
    # N > 1 line from N processes on ONE GPU over a test double of the nccl* calls must not be able to pass for a scaling point
    if hx.rccl is not None:
        rccl_library, rccl_override, bus = hx.rccl.transport_info()
    else:
        rccl_library, rccl_override, bus = None, False, device_bus_id(dev)
    buses = [None] * world
    dist.all_gather_object(buses, "%s" % (bus,))
    distinct_devices = len(set(buses))
    # correctness outside the timed region: features that are a closed form of the GLOBAL row id, so every rank can check
    # the halo rows it pulled and (on its first rows) the aggregation against the oracle without any further exchange

    def closed_form(ids):
        r = torch.as_tensor(ids, dtype=torch.int64, device=dev)[:, None]
        c = torch.arange(feat, dtype=torch.int64, device=dev)[None, :]
        return (((r * 131 + c * 71) % 1013).to(torch.float32) / 1013.0 - 0.5)
    pg.set_local_x(closed_form(np.arange(r0, r1)))
    dog.arm(120, "first step over the %s transport" % transport)
    y_chk = pg.step().clone()
    torch.cuda.synchronize()
    dog.disarm()
    ok = bool(torch.equal(pg.x_halo, closed_form(hx.halo_ids))) if hx.n_halo else True
    from oracle import oracle as orc
    nchk = min(hx.n_local, 2000)
    lp = hx.local_ptr[:nchk + 1]
    x_ext = pg.x_ext.cpu().numpy()
    ref = orc.gcn_seq(lp, hx.local_idx[:lp[-1]], val_slice[:lp[-1]], x_ext)
    scale = orc.gcn_abs_scale(lp, hx.local_idx[:lp[-1]], val_slice[:lp[-1]], x_ext)
    ok = ok and bool(np.all(np.abs(y_chk[:nchk].cpu().numpy() - ref) <= 1e-5 * scale + 1e-30))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        log("rank %d: row-partitioned step over the %s transport differs from the oracle / halo rows differ from their owners' rows (this rank ok: %s)"
            % (rank, transport, ok))
        dist.barrier()
        os._exit(17)   # every rank takes this exit together; the supervisor starts the fallback transport in fresh processes
    g = torch.Generator(device=dev)
    g.manual_seed(123 + rank)
    pg.set_local_x(torch.randn((hx.n_local, feat), generator=g, device=dev))
    dog.arm(120 + 0.5 * (steps + warmup) * 3, "timed steps over the %s transport" % transport)
    wall, dev_s, med_s = time_steps(pg.step, steps, warmup, dist.barrier)
    dog.disarm()
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    halo = torch.tensor([hx.halo_bytes(feat), pg.num_e_local, pg.num_e_remote if pg.overlap else 0], dtype=torch.float64, device=dev)
    halo_max = halo.clone()
    dist.all_reduce(halo, op=dist.ReduceOp.SUM)
    dist.all_reduce(halo_max, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    # Context, reported beside the line and never as `value`: the same kernels with the halo rows already resident (static
    # input features, i.e. the exchange hoisted out of the step) -- what the step costs when the xGMI exchange is free.
    wall_nx, _, _ = time_steps(lambda: pg.compute("sum", None), steps, warmup, dist.barrier)
    t_nx = torch.tensor([wall_nx], dtype=torch.float64, device=dev)
    dist.all_reduce(t_nx, op=dist.ReduceOp.MAX)
    wall_nx = float(t_nx.item())
    # The parts of the step alone, so that an N > 1 line explains itself on first contact with hardware: the exchange
    # (pack kernel + the step's own all-to-all-v(s): same transport, same per-peer byte counts, nothing beside it on the device), the
    # local-source pass, the halo-source passes.  predicted = max(exchange, local) + halo: what the overlap plan should cost if the exchange
    # and the local pass do not slow each other down.
    dog.arm(120 + 0.5 * (steps + warmup) * 3, "exchange alone over the %s transport" % transport)
    parts = {}
    parts["exchange_alone"], _, _ = time_steps(lambda: hx.exchange(pg.x_local, pg.x_halo, pg.send_buf, async_op=False), steps, warmup, dist.barrier)
    dog.disarm()
    if pg.overlap:
        parts["local_pass"], _, _ = time_steps(lambda: pg.agg_loc.run(pg.x_local, pg.y, 512, "balanced"), steps, warmup, dist.barrier)

        def halo_passes():
            for a in pg.agg_rem_stages:
                if a is not None:
                    a.run(pg.x_halo, pg.y, 512, "balanced", accumulate=True)
        parts["halo_pass"], _, _ = time_steps(halo_passes, steps, warmup, dist.barrier)
    else:
        parts["local_pass"], parts["halo_pass"] = wall_nx, 0.0
    tparts = torch.tensor([parts["exchange_alone"], parts["local_pass"], parts["halo_pass"]], dtype=torch.float64, device=dev)
    dist.all_reduce(tparts, op=dist.ReduceOp.MAX)
    peer_rows = torch.tensor([float(max([int(v) for p_, v in enumerate(hx.recv_counts) if p_ != rank] or [0]))], dtype=torch.float64, device=dev)
    dist.all_reduce(peer_rows, op=dist.ReduceOp.MAX)
    tp = torch.tensor([t_plan], dtype=torch.float64, device=dev)
    dist.all_reduce(tp, op=dist.ReduceOp.MAX)
    n_stages, stage_mode = hx.n_stages, "%s x %d" % (hx.stage_mode, hx.n_stages)
    pg.close()
    if hx.rccl is not None:
        hx.rccl.close()
    del pg
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    return {"Vg": Vg, "Eg": Eg, "feat": feat, "step_s": wall / steps, "step_nx_s": wall_nx / steps, "steps": steps, "warmup": warmup,
            "halo_bytes_all": float(halo[0].item()), "halo_bytes_max_rank": float(halo_max[0].item()),
            "remote_edge_share": float(halo[2].item()) / max(float(halo[1].item()), 1.0),
            "rccl_ranks": rccl_ranks, "plan_s": float(tp.item()), "n_stages": n_stages, "stage_mode": stage_mode,
            "rccl_library": rccl_library, "rccl_library_is_override": rccl_override, "distinct_devices": distinct_devices,
            "device_pci_bus_ids": buses,
            "exchange_alone_s": float(tparts[0].item()) / steps, "local_pass_s": float(tparts[1].item()) / steps,
            "halo_pass_s": float(tparts[2].item()) / steps, "largest_message_bytes": float(peer_rows.item()) * feat * 4}


def device_bus_id(dev):
    """PCI bus id of a torch device (the torch transports have no C-ABI communicator to ask)."""
    import ctypes
    buf = ctypes.create_string_buffer(64)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev.index or 0)) == 0:
            return buf.value.decode()
    except OSError:
        pass
    return "device-%d" % int(dev.index or 0)


TRANSPORT_NAMES = {"rccl": "C-ABI step (gnnagg_dist_step_gcn: pack kernel + grouped ncclSend/ncclRecv per stage on the step's own stream)",
                   "torch": "torch.distributed all_to_all_single per stage + the aggregation launches from Python"}


def run_multi(args, dev, rank, world, dog):
    import torch.distributed as dist
    backend = dist.get_backend()
    transport = os.environ.get("BENCH_TRANSPORT") or ("rccl" if backend == "nccl" else "torch")
    strong = args.config == "P"  # BASELINE configs[4]: ONE products-shaped graph row-partitioned over the N GPUs
    m = partitioned_run(args, dev, rank, world, strong, args.steps, args.warmup, transport, dog)
    # one driver pass yields both: the N > 1 headline line (weak scaling, feat 128) carries BASELINE configs[4] -- the products-shaped
    # graph strong-scaled over the same ranks -- as a sub-record (fewer steps: a step is milliseconds there)
    ps = None
    if not strong and os.environ.get("BENCH_PRODUCTS", "1") != "0":
        ps = partitioned_run(args, dev, rank, world, True, max(2, min(args.steps, 10)), min(args.warmup, 2), transport, dog)
    if rank != 0:
        return None
    feat, Vg, Eg, step_s = m["feat"], m["Vg"], m["Eg"], m["step_s"]
    B = algorithmic_bytes(Vg, Eg, feat)
    double = bool(m["rccl_library_is_override"]) or m["distinct_devices"] < world
    if double:
        how = []
        if m["distinct_devices"] < world:
            how.append("%d ranks on %d GPU%s" % (world, m["distinct_devices"], "" if m["distinct_devices"] == 1 else "s"))
        if m["rccl_library_is_override"]:
            how.append("a test double of the nccl* entry points (GNNAGG_RCCL_LIB = %s)" % m["rccl_library"])
        link = "halo pull per step -- FUNCTIONAL CHECK, NOT A SCALING POINT: " + ", ".join(how) + "; no interconnect was exercised"
    else:
        link = "halo pull per step over xGMI"
    transport_is = TRANSPORT_NAMES[transport] + ("" if not double else " -- " + link)

    def parts_of(mm):
        """exchange / passes alone beside the step (all max over ranks): predicted_step_ms = max(exchange, local) + halo; link_gbps_per_peer =
        the largest pairwise message of the step / the exchange's time (every pairwise message rides its own xGMI link, so the largest
        one sets the time when the links are the limit)"""
        ex, lo, ha = mm["exchange_alone_s"] * 1e3, mm["local_pass_s"] * 1e3, mm["halo_pass_s"] * 1e3
        return {"exchange_alone_ms": ex, "local_pass_ms": lo, "halo_pass_ms": ha, "predicted_step_ms": max(ex, lo) + ha,
                "link_gbps_per_peer": (mm["largest_message_bytes"] / mm["exchange_alone_s"] / 1e9) if mm["exchange_alone_s"] > 0 else None,
                "largest_message_bytes": mm["largest_message_bytes"]}
    out = {
        "metric": "aggregated edges/sec, GCN SpMM feat=%d" % feat, "value": Eg / step_s, "unit": "edges/s",
        # n_gpus = the GPUs the ranks really sit on: `ranks` processes on ONE device (a functional check on a one-GPU box) is not an N-GPU point
        "n_gpus": m["distinct_devices"], "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s CSR %dx%d (seed 123, generated in its community order = a perfect locality reorder; %d %% global picks), GCN sum, "
                               "feat=%d, 1-D row partition + %s; stages: %s" % (
                                   "products-shaped" if strong else "%d x arxiv-shaped" % world, Vg, Eg, round(args.global_share * 100), feat,
                                   link, m["stage_mode"]),
                   "num_v": Vg, "num_e": Eg, "feat": feat, "halo_bytes_per_step_all_ranks": m["halo_bytes_all"],
                   "global_share": args.global_share, "verified_against_oracle": True,
                   "graph_numbering": "generator's hidden community order (perfect-reorder assumption: 73.2 vs 73.7 us on A, DESIGN.md 6)"},
        "transport": transport, "transport_is": transport_is, "backend": backend, "rccl_ranks": m["rccl_ranks"],
        "rccl_library": m["rccl_library"], "rccl_library_is_override": m["rccl_library_is_override"],
        "distinct_devices": m["distinct_devices"], "device_pci_bus_ids": m["device_pci_bus_ids"], "test_double": double,
        "transport_fallback": os.environ.get("BENCH_FALLBACK_REASON"),
        "halo_stages": m["n_stages"], "plan_s": m["plan_s"],
        "achieved_gbps": B / step_s / 1e9,
        # halo bytes (config.halo_bytes_per_step_all_ranks) and the exposed communication time = what the step costs
        # beyond the same kernels with the halo rows already resident
        "exposed_comm_ms_per_step": max(0.0, (m["step_s"] - m["step_nx_s"]) * 1e3),
        "remote_edge_share": m["remote_edge_share"],
        "no_exchange_upper_bound": {"value": Eg / m["step_nx_s"], "ms_per_step": m["step_nx_s"] * 1e3},
        "step_parts": parts_of(m),
        # per-GPU share of the step, halo exchange included: the bound is whichever of the xGMI links and the memory system
        # is slower for this partition -- reported against the HBM figure for continuity with the 1-GPU line, not as a
        # kernel roofline (that is the N = 1 line's job)
        "roofline": {"bound": "hbm", "achieved": B / step_s / 1e9 / world, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": B / step_s / 1e9 / world / HBM_PEAK_GBPS, "traffic": None,
                     "frac_is": "per-rank share of the gather-model bytes / step time (exchange included) / 8 TB/s; DESIGN.md 5",
                     "kernel": "per-rank share of the whole step (halo exchange included)", "algorithmic_bytes": B,
                     "halo_bytes_per_rank": m["halo_bytes_all"] / world, "halo_bytes_max_rank": m["halo_bytes_max_rank"]},
    }
    if ps is not None:
        out["products_strong"] = {
            "what": "BASELINE configs[4]: ONE products-shaped CSR %dx%d, feat=100, GCN sum, over the same %d ranks (strong scaling)" % (ps["Vg"], ps["Eg"], world),
            "step_parts": parts_of(ps),
            "value": ps["Eg"] / ps["step_s"], "unit": "edges/s", "ms_per_step": ps["step_s"] * 1e3, "steps": ps["steps"], "warmup": ps["warmup"],
            "scaling": "strong", "halo_bytes_per_step_all_ranks": ps["halo_bytes_all"], "halo_bytes_max_rank": ps["halo_bytes_max_rank"],
            "exposed_comm_ms_per_step": max(0.0, (ps["step_s"] - ps["step_nx_s"]) * 1e3),
            "no_exchange_ms_per_step": ps["step_nx_s"] * 1e3, "remote_edge_share": ps["remote_edge_share"], "halo_stages": ps["n_stages"],
            "plan_s": ps["plan_s"], "rccl_ranks": ps["rccl_ranks"], "verified_against_oracle": True,
            "distinct_devices": ps["distinct_devices"], "rccl_library": ps["rccl_library"],
            "test_double": bool(ps["rccl_library_is_override"]) or ps["distinct_devices"] < world}
    sp = out["step_parts"]
    out["summary"] = {"ranks": world, "distinct_devices": m["distinct_devices"], "test_double": double, "transport": transport, "stages": m["stage_mode"],
                      "ms_per_step": round(step_s * 1e3, 4), "edges_per_s": round(Eg / step_s), "no_exchange_ms": round(m["step_nx_s"] * 1e3, 4),
                      "exchange_alone_ms": round(sp["exchange_alone_ms"], 4), "local_pass_ms": round(sp["local_pass_ms"], 4),
                      "halo_pass_ms": round(sp["halo_pass_ms"], 4), "predicted_step_ms": round(sp["predicted_step_ms"], 4),
                      "link_gbps_per_peer": None if sp["link_gbps_per_peer"] is None else round(sp["link_gbps_per_peer"], 2),
                      "halo_mb_per_rank": round(m["halo_bytes_all"] / world / 1e6, 2), "verified": True}
    if ps is not None:
        q = out["products_strong"]["step_parts"]
        out["summary"]["P"] = {"ms_per_step": round(ps["step_s"] * 1e3, 4), "edges_per_s": round(ps["Eg"] / ps["step_s"]),
                               "exchange_alone_ms": round(q["exchange_alone_ms"], 4), "local_pass_ms": round(q["local_pass_ms"], 4),
                               "halo_pass_ms": round(q["halo_pass_ms"], 4), "predicted_step_ms": round(q["predicted_step_ms"], 4),
                               "stages": ps["stage_mode"], "halo_mb_max_rank": round(ps["halo_bytes_max_rank"] / 1e6, 2)}
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relay_json(stdout_bytes, json_fd):
    """writes the child's JSON line(s) to the real stdout; returns how many there were"""
    n = 0
    for line in stdout_bytes.decode(errors="replace").splitlines():
        if line.strip().startswith("{"):
            os.write(json_fd, (line.strip() + "\n").encode())
            n += 1
    return n


def launch_ranks(args, json_fd):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes through torch.distributed.run.  This process has not
    imported torch, let alone touched a GPU."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: no launcher environment, starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE)
    n = relay_json(r.stdout, json_fd)
    return r.returncode if r.returncode != 0 else (0 if n == 1 else 1)


def sup_dir():
    """Directory the supervisors of ONE job on ONE node share.  Every supervisor of a node is a child of the same launcher process, so
    the name carries that parent's pid AND its start time (field 22 of /proc/<pid>/stat: a later launcher that happens to get the same pid
    cannot collide, so nothing ever has to be removed before use) and the job's rendezvous port."""
    import tempfile
    ppid = os.getppid()
    try:
        start = open("/proc/%d/stat" % ppid).read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        start = "0"
    return os.path.join(tempfile.gettempdir(), "gnnagg_bench_sup_%d_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0")))


def agree_on_outcome(local_rank, local_world, attempt, rc, timeout_s=420.0, need_port=True):
    """The supervisors of ONE NODE agree on what an attempt did before any of them starts the next one (a rank whose child
    failed must not open a rendezvous that the ranks whose children returned 0 never join).  The file set is the node's own ranks
    (LOCAL_RANK / LOCAL_WORLD_SIZE: the directory is node-local -- ADVICE r5); the local leader also drops the port of the next
    rendezvous (a free one, checked).  Returns (worst exit code over the node's ranks, next port), or (None, None) when some rank
    never reported -- then nobody retries.  Multi-node jobs do not use it (supervise_rank: no fallback ladder there)."""
    d = os.path.join(sup_dir(), "attempt%d" % attempt)
    os.makedirs(d, exist_ok=True)

    def drop(name, text):
        tmp = os.path.join(d, ".%s.%d" % (name, os.getpid()))
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, os.path.join(d, name))
    if local_rank == 0 and need_port:
        drop("port", str(free_port()))
    drop("rank%d" % local_rank, str(rc))
    t_end = time.monotonic() + timeout_s
    names = ["rank%d" % r for r in range(local_world)] + (["port"] if need_port else [])
    while time.monotonic() < t_end:
        if all(os.path.exists(os.path.join(d, n)) for n in names):
            codes = [int(open(os.path.join(d, "rank%d" % r)).read().strip() or "1") for r in range(local_world)]
            worst = next((c for c in codes if c != 0), 0)
            return worst, (int(open(os.path.join(d, "port")).read().strip()) if need_port else None)
        time.sleep(0.2)
    return None, None


def supervise_rank(args, json_fd):
    """A rank process under a launcher (WORLD_SIZE > 1): runs the real rank in a FRESH child -- first on the C-ABI RCCL step, and
    if any rank's child exits non-zero (oracle mismatch: 17, watchdog: 18, a crash) once more in another fresh child on
    torch.distributed's all_to_all_single.  The decision is taken by all supervisors together (agree_on_outcome).  Never touches the
    GPU itself; nothing is restarted in place."""
    import subprocess
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    first = os.environ.get("BENCH_TRANSPORT") or ("rccl" if backend == "nccl" else "torch")
    # (transport, backend) in the order tried: the C-ABI RCCL step; torch.distributed's all_to_all_single over the same RCCL; and, when
    # the nccl backend itself is what fails (no peer access, a broken fabric), the same all_to_all_single over gloo -- halo rows
    # staged through the host: slow, still a measured and oracle-checked line, and it says which transport it is
    attempts = [(first, backend)]
    if os.environ.get("BENCH_NO_FALLBACK") != "1":
        if first == "rccl":
            attempts.append(("torch", backend))
        if backend == "nccl":
            attempts.append(("torch", "gloo"))
    # the agreement is per node (its files are node-local).  A job that spans nodes has no channel here to agree across them, so it runs
    # the first transport only and every supervisor reports its own child (the global WORLD_SIZE never fits a node's directory)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    single_node = local_world == world
    if not single_node:
        attempts = attempts[:1]
    rc, reasons, port = 1, [], None
    for i, (tr, be) in enumerate(attempts):
        env = dict(os.environ, BENCH_CHILD="1", BENCH_TRANSPORT=tr, BENCH_BACKEND=be)
        if i > 0:
            # a fresh rendezvous for the fresh processes: rank 0's child hosts the store itself, on the port rank 0's supervisor picked
            env["MASTER_PORT"] = str(port)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            reasons.append("the %s transport's ranks (backend %s) exited with code %d" % (attempts[i - 1][0], attempts[i - 1][1], rc))
            env["BENCH_FALLBACK_REASON"] = "; ".join(reasons) + " (17: first step failed the oracle check, 18: watchdog on a hung exchange or " \
                                           "rendezvous); this line was measured on the fallback transport in fresh processes"
            log("bench.py rank %d: %s" % (rank, env["BENCH_FALLBACK_REASON"]))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE)
        own = r.returncode
        if not single_node:
            if rank == 0 and own == 0:
                return 0 if relay_json(r.stdout, json_fd) == 1 else 1
            return own
        last = i + 1 == len(attempts)
        rc, port = agree_on_outcome(local_rank, local_world, i, own, need_port=not last)
        if rc is None:
            # some supervisor never reported: nobody starts another attempt.  A rank whose own child succeeded says so (its line, if it is
            # rank 0, is still the measurement); only a failed child makes this supervisor fail
            log("bench.py rank %d: the other ranks' supervisors never reported attempt %d; no further attempt" % (rank, i))
            if own == 0 and rank == 0:
                return 0 if relay_json(r.stdout, json_fd) == 1 else 1
            return own
        if rc == 0:
            if rank == 0:
                return 0 if relay_json(r.stdout, json_fd) == 1 else 1
            return 0
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    ap.add_argument("--config", default="A", choices=["A", "R", "G", "P1", "P"],
                    help="A (default): arxiv-shaped GCN sum feat=128, the headline line (weak scaling for N > 1, with the products-shaped "
                         "strong-scaling sub-record); R/G/P1: the other 1-GPU configs; P (N > 1): products-shaped GCN feat=100, strong scaling")
    ap.add_argument("--global-share", type=float, default=0.5,
                    help="N > 1: share of a row's sources the generator draws from the global popularity distribution (the rest come from a "
                         "window around the row: what a row partition can keep local).  Default 0.5 = the generator every other line uses")
    args = ap.parse_args()
    if not 0.0 <= args.global_share <= 1.0:
        raise SystemExit("--global-share must be in [0, 1]")

    # stdout carries exactly ONE JSON line.  RCCL prints a version banner through C stdio that is flushed at
    # process exit, so fd 1 is pointed at stderr for the whole run and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # N > 1: the processes that touch GPUs are always FRESH children (see the module docstring); none of the two functions below
    # imports torch or loads the HIP library
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        rc = launch_ranks(args, json_fd)
        os.close(json_fd)
        sys.exit(rc)
    if launched and int(os.environ["WORLD_SIZE"]) > 1 and os.environ.get("BENCH_CHILD") != "1":
        rc = supervise_rank(args, json_fd)
        os.close(json_fd)
        sys.exit(rc)

    global np, torch
    import numpy as np
    import torch

    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "gnn_computing_amd", "libgnnagg.so")):
        ge.build()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # test hooks (not used by the driver): BENCH_ONE_GPU=1 puts every rank on cuda:0 and BENCH_BACKEND=gloo swaps the
    # transport, so the N > 1 code path can be exercised on a single-GPU box (RCCL needs one GPU per rank)
    if os.environ.get("BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("BENCH_FORCE_MULTI") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        dog = Watchdog()
        dog.arm(300, "rendezvous + communicator (%s backend)" % backend)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        out = run_multi(args, dev, rank, world, dog)
        dist.barrier()
        dist.destroy_process_group()
    else:
        if args.gpus != 1:
            raise SystemExit("--gpus %d: WORLD_SIZE is 1 in the environment" % args.gpus)
        if args.config == "P":
            args.config = "P1"
        out = run_single(args, dev) if args.config == "A" else run_other_config(args, dev)
    if rank == 0 and out is not None:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
