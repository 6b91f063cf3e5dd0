"""bench.py's output contract, exercised the way the driver runs it (one JSON line on stdout, the fields the round's
instructions name, roofline.frac tied to a hardware peak, the oracle check inside bench.py passing) -- short runs of the
headline line, of one other config, and of the N > 1 code paths (weak arxiv-shaped line, strong products-shaped --config P) with
two ranks sharing the GPU over gloo (test hooks of bench.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def run(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    if r.returncode != 0 and launcher is not None:
        # the multi-process lines rendezvous over TCP on a box that was booted seconds ago; one such launch in ~60 has failed without
        # reproducing.  ONE more attempt on another port, with the first failure printed -- a real defect fails twice
        import warnings
        warnings.warn("bench.py under a launcher exited %d, retrying once on another port; stderr tail:\n%s" % (r.returncode, r.stderr[-3000:]))
        cmd = [str(int(c) + 17) if i > 0 and cmd[i - 1] == "--master-port" else c for i, c in enumerate(cmd)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line, got %d" % len(lines)
    check_summary_tail(lines[0])
    return json.loads(lines[0])


def check_summary_tail(line):
    """the driver keeps 2 000 characters of the line's tail and truncates strings -- so the line ENDS with a compact
    `summary` object (<= 1 KB) that carries every record's numbers, and no string in the line is a paragraph."""
    d = json.loads(line)
    assert list(d)[-1] == "summary", list(d)[-3:]
    tail = json.dumps(d["summary"])
    assert len(tail) <= 1024 and line.rstrip().endswith(tail + "}"), len(tail)

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    long = [t for t in strings(d) if len(t) > 200 and not t.startswith("/")]
    # (the N > 1 workload / fallback strings say what carried the halo rows and why: they may run to a few hundred characters)
    assert all(len(t) <= 700 for t in long) and len(long) <= 4, [t[:80] for t in long]


def check_common(d, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    # N > 1 lines from ranks that share ONE device say n_gpus = distinct devices and carry the rank count in `ranks`
    assert d.get("ranks", d["n_gpus"]) == n_gpus and d["n_gpus"] == d.get("distinct_devices", n_gpus) and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "edges/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # frac is counter traffic (or gather-model bytes) over a HARDWARE peak, never a probe of the same kernel, and is not clamped:
    # cache-served gather bytes may exceed the HBM figure -- that is reported, not hidden
    assert 0.0 < r["frac"] < 2.0 and r["peak"] in (8000.0, 34500.0) and r["bound"] in ("hbm", "l2")
    assert d["value"] > 0 and abs(d["value"] - d["config"]["num_e"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.05 or n_gpus > 1


def test_headline_line():
    d = run(["--steps", "20", "--warmup", "3", "--cpu-budget", "1"])
    check_common(d, 1, 20, 3)
    assert d["metric"].startswith("aggregated edges/sec") and d["config"]["num_v"] == 169343 and d["config"]["num_e"] == 1166243
    assert d["config"]["feat"] == 128
    r = d["roofline"]
    assert r["algorithmic_bytes"] == 693827352 and r["compulsory_bytes"] == 183414552 and r["ceiling_probe_us"] > 0
    assert r["peak"] == 8000.0 and abs(r["achieved"] - 693827352 / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert 0.5 < r["probe_frac"] < 1.2 and "traffic_stale" in r
    if r["traffic"]:   # frac = counter traffic / time / 8 TB/s, recomputable from the line
        assert abs(r["frac"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "edges/s" and c["sample"]
    assert d["value"] > 8.07e9          # north_star: >= 60 % of the HBM-read roofline in gather-model bytes = 8.07 G edges/s
    # a denominator that bounds the bytes it divides: the ceilings are measured in the bench process
    assert abs(r["frac_vs_gather_ceiling"] - r["achieved"] / r["gather_ceiling"]["gbps"]) < 1e-9 and "Infinity-Cache" in r["frac_is"]
    assert 4000 < r["gather_ceiling_hbm"]["gbps"] < 9000 and 5000 < r["gather_ceiling"]["gbps"] < 12000
    # every other single-GPU configuration of BASELINE.json rides in the same driver command, oracle-verified
    sizes = {"A_rows": (169343, 1166243, 128), "R": (232965, 114615891, 602), "G": (232965, 114615891, 256), "P1": (2449029, 123718280, 100)}
    assert set(d["configs"]) == set(sizes)
    for name, (V, E, F) in sizes.items():
        c = d["configs"][name]
        assert "error" not in c, (name, c)
        assert (c["config"]["num_v"], c["config"]["num_e"], c["config"]["feat"]) == (V, E, F)
        assert c["verified_against_oracle"] is True and c["verified_rows"] > 0 and c["steps"] >= 1
        assert abs(c["value"] - E / (c["ms_per_step"] * 1e-3)) < 1e-6 * c["value"]
        q = c["roofline"]
        assert q["avg_launch_us"] > 0 and q["algorithmic_bytes"] > 0 and "frac_is" in q and "traffic" in q and "probe_frac" in q
        assert abs(q["achieved"] - q["algorithmic_bytes"] / (q["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * q["achieved"]
        assert abs(q["frac_vs_gather_ceiling"] - q["achieved"] / q["gather_ceiling"]["gbps"]) < 1e-9
        assert "schedule_prep_s" in c and c["scratch_bytes"] >= 0
        if name != "A_rows":
            assert c["worst_ratio_to_1e-5_bound"] <= 1.0
    assert d["configs"]["R"]["roofline"]["bound"] == "l2" and d["configs"]["P1"]["roofline"]["bound"] == "hbm"
    assert d["configs"]["A_rows"]["value"] < d["value"]     # the canonical chains cost more than the balanced order
    # P1 is timed with the locality reorder applied on load, like the headline; the generator's numbering rides beside it
    p1 = d["configs"]["P1"]
    assert p1["with_locality_reorder"]["verified_against_oracle"] is True and "reorder" in p1["config"]["workload"]
    assert p1["no_reorder"]["verified_against_oracle"] is True and p1["no_reorder"]["value"] > 0 and p1["no_reorder"]["worst_ratio_to_1e-5_bound"] <= 1.0
    # the summary the driver's tail shows: every record's [ms_per_step, frac, frac_vs_gather_ceiling, verified_rows]
    sm = d["summary"]
    assert sm["verified"] is True and set(sm) >= {"A", "A_rows", "R", "G", "P1", "P1_no_reorder", "edges_per_s", "cpu_edges_per_s"}
    for name in ("A_rows", "R", "G", "P1"):
        assert abs(sm[name][0] - d["configs"][name]["ms_per_step"]) < 1e-4 and sm[name][3] == d["configs"][name]["verified_rows"]
    assert abs(sm["A"][0] - d["ms_per_step"]) < 1e-4 and abs(sm["A"][1] - r["frac"]) < 1e-3


def test_other_config_line():
    # (BENCH_P1_REORDER=0: the reordered arm -- 40 s of generator on a fresh box -- is timed and verified by test_headline_line's P1 sub-record)
    d = run(["--config", "P1", "--steps", "3", "--warmup", "1", "--no-cpu"], env={"BENCH_P1_REORDER": "0"})
    check_common(d, 1, 3, 1)
    assert d["config"]["feat"] == 100 and d["roofline"]["ceiling_probe_us"] > 0


def check_multi(d):
    assert d["scaling"] == "weak" and d["config"]["verified_against_oracle"] is True and d["config"]["num_e"] == 2 * 1166243
    assert d["no_exchange_upper_bound"]["value"] >= d["value"] * 0.9
    assert d["transport"] in ("rccl", "torch") and "rccl_ranks" in d and d["halo_stages"] >= 1 and d["config"]["global_share"] == 0.5
    p = d["products_strong"]      # ONE pass yields the feat-128 line and BASELINE configs[4] over the same ranks
    assert p["scaling"] == "strong" and p["value"] > 0 and p["verified_against_oracle"] is True and p["halo_bytes_per_step_all_ranks"] > 0
    assert "123718280" in p["what"] and p["exposed_comm_ms_per_step"] >= 0
    # the parts of the step alone, so that the line explains itself (values mean nothing on one GPU, the schema does)
    for rec in (d, p):
        q = rec["step_parts"]
        assert q["exchange_alone_ms"] > 0 and q["local_pass_ms"] > 0 and q["halo_pass_ms"] >= 0 and q["link_gbps_per_peer"] > 0
        assert abs(q["predicted_step_ms"] - (max(q["exchange_alone_ms"], q["local_pass_ms"]) + q["halo_pass_ms"])) < 1e-9
    sm = d["summary"]
    assert sm["ranks"] == d["ranks"] and abs(sm["ms_per_step"] - d["ms_per_step"]) < 1e-3 and "P" in sm and sm["verified"] is True


def test_two_ranks_on_one_gpu_over_gloo():
    """the driver's N > 1 command: torch.distributed.run starts the ranks (each a supervisor + a fresh child)"""
    d = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"], env={"BENCH_ONE_GPU": "1", "BENCH_BACKEND": "gloo"},
            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29571"])
    check_common(d, 2, 3, 1)
    check_multi(d)
    assert d["transport"] == "torch" and d["rccl_ranks"] is None and d["transport_fallback"] is None
    # two ranks on ONE GPU: the line must say so and never claim an interconnect
    assert d["distinct_devices"] == 1 and d["test_double"] is True and "NOT A SCALING POINT" in d["config"]["workload"]
    assert "xGMI" not in d["config"]["workload"] and "xGMI" not in d["transport_is"] and d["products_strong"]["test_double"] is True


def test_two_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2` the way the driver launches N = 1: no launcher, WORLD_SIZE unset.  bench.py starts the ranks
    itself in fresh child processes (before anything touches the GPU) and relays their one line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_ONE_GPU="1", BENCH_BACKEND="gloo", BENCH_PRODUCTS="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", "--global-share", "0.25"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_common(d, 2, 3, 1)
    assert d["config"]["global_share"] == 0.25 and "products_strong" not in d and d["remote_edge_share"] < 0.2   # the locality knob


def test_failed_rccl_ranks_fall_back_to_torch_transport_in_fresh_processes():
    """The C-ABI RCCL step is the default transport with the nccl backend; when its ranks fail -- here: two ranks on ONE GPU, which
    RCCL refuses -- every supervisor starts the rank again in a fresh child on all_to_all_single, with a fresh rendezvous, and the
    line says so.  The same path catches a failed oracle check (exit 17) and a hung exchange (watchdog, 18)."""
    d = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"],
            env={"BENCH_ONE_GPU": "1", "BENCH_BACKEND": "gloo", "BENCH_TRANSPORT": "rccl", "BENCH_PRODUCTS": "0"},
            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29575"])
    check_common(d, 2, 2, 1)
    assert d["transport"] == "torch" and d["transport_fallback"] and "rccl" in d["transport_fallback"]


def check_labelled_as_double(d, fake):
    """a record that says n_gpus = N, rccl_ranks = N from ONE GPU must not be able to pass for a scaling point -- the
    library names the file its nccl* entry points came from (gnnagg_dist_transport_info), the ranks all-gather their PCI bus ids."""
    assert os.path.samefile(d["rccl_library"], fake) and d["rccl_library_is_override"] is True
    assert d["distinct_devices"] == 1 and d["n_gpus"] == 1 and len(set(d["device_pci_bus_ids"])) == 1 and len(d["device_pci_bus_ids"]) == d["ranks"]
    assert d["test_double"] is True and "NOT A SCALING POINT" in d["config"]["workload"] and "test double" in d["transport_is"]
    assert "over xGMI" not in d["config"]["workload"] and "over xGMI" not in d["transport_is"]


def test_two_ranks_on_the_cabi_rccl_step_through_the_test_double():
    """The default transport of the N > 1 line -- the one-call C-ABI step, gnnagg_dist_step_gcn -- through bench.py's own code path
    (communicator from gnnagg_dist_comm_create, `rccl_ranks` from gnnagg_dist_comm_info, oracle check of the first step, timed
    steps) with two ranks on the one GPU: the nccl entry points are served by tests/fake_rccl (GNNAGG_RCCL_LIB), the plan
    exchange by gloo.  No fallback may be taken."""
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
    assert os.path.exists(fake), "tests/fake_rccl/libfakerccl.so is not built (__graft_entry__.build() builds it)"
    d = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"],
            env={"BENCH_ONE_GPU": "1", "BENCH_BACKEND": "gloo", "BENCH_TRANSPORT": "rccl", "BENCH_PRODUCTS": "0", "BENCH_NO_FALLBACK": "1",
                 "GNNAGG_RCCL_LIB": fake},
            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29577"])
    check_common(d, 2, 3, 1)
    assert d["transport"] == "rccl" and d["rccl_ranks"] == 2 and d["transport_fallback"] is None and d["halo_stages"] >= 1
    assert d["config"]["verified_against_oracle"] is True
    check_labelled_as_double(d, fake)


def test_eight_ranks_on_the_cabi_rccl_step_through_the_test_double():
    """The rank count the line is specified for (BASELINE: 1 / 2 / 4 / 8 GPUs): `python3 bench.py --gpus 8` without a launcher, eight
    ranks on the one GPU, the one-call C-ABI step with seven peers per rank and the owner plan's seven stages (one ring distance each) --
    the nccl entry points served by tests/fake_rccl.  Every rank's first step is checked against the oracle inside bench.py."""
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
    assert os.path.exists(fake), "tests/fake_rccl/libfakerccl.so is not built (__graft_entry__.build() builds it)"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_ONE_GPU="1", BENCH_BACKEND="gloo", BENCH_TRANSPORT="rccl", BENCH_PRODUCTS="0", BENCH_NO_FALLBACK="1", BENCH_STAGES="owner",
               GNNAGG_RCCL_LIB=fake)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_common(d, 8, 2, 1)
    assert d["transport"] == "rccl" and d["rccl_ranks"] == 8 and d["halo_stages"] == 7 and d["transport_fallback"] is None
    assert d["config"]["verified_against_oracle"] is True and d["config"]["num_e"] == 8 * 1166243
    assert 0.3 < d["remote_edge_share"] < 0.5      # 7/8 of the generator's 50 % global picks
    check_labelled_as_double(d, fake)


@pytest.mark.skipif(os.environ.get("GNNAGG_TEST_TIER") != "2", reason="second tier (18 s): the ladder's first rung runs in every pass "
                    "(test_failed_rccl_ranks_fall_back_to_torch_transport_in_fresh_processes)")
def test_failed_nccl_backend_falls_back_to_gloo_in_fresh_processes():
    """Third level of the same ladder: when the nccl backend itself fails -- here: the driver's exact N = 2 launch with both ranks on ONE
    GPU, which RCCL refuses for the C-ABI step AND for torch.distributed -- the ranks start once more on all_to_all_single over gloo
    (halo rows through the host).  A slow line that says what it is instead of rc != 0."""
    d = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"], env={"BENCH_ONE_GPU": "1", "BENCH_PRODUCTS": "0"},
            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29581"])
    check_common(d, 2, 2, 1)
    assert d["transport"] == "torch" and d["backend"] == "gloo" and d["rccl_ranks"] is None
    assert "rccl" in d["transport_fallback"] and "backend nccl" in d["transport_fallback"]


def test_products_strong_scaling_two_ranks_on_one_gpu_over_gloo():
    """BASELINE configs[4] through the N > 1 code: ONE products-shaped graph (2 449 029 x 123 718 280, feat 100) row-partitioned over
    the ranks, halo pull inside every timed step, every rank checked against the oracle before the timed region."""
    d = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--config", "P"], env={"BENCH_ONE_GPU": "1", "BENCH_BACKEND": "gloo"},
            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", "29573"])
    check_common(d, 2, 2, 1)
    assert d["scaling"] == "strong" and d["config"]["verified_against_oracle"] is True
    assert d["config"]["num_v"] == 2449029 and d["config"]["num_e"] == 123718280 and d["config"]["feat"] == 100
    assert d["config"]["halo_bytes_per_step_all_ranks"] > 0
