"""CPU: the host logic of bench.py that needs no GPU -- the supervisors' agreement before a transport fallback, the gather-model byte
counts of SURVEY 8d, and the float64 yardstick the sub-records are verified against."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import bench
rank, world, rc = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out = []
for attempt in range(2):
    out.append(bench.agree_on_outcome(rank, world, attempt, rc if attempt == 0 else 0, timeout_s=30.0))
print(json.dumps(out))
"""


def test_supervisors_agree_before_falling_back():
    """a subset of ranks failing must not leave the failed ranks alone in a new rendezvous.  Three supervisors (children of
    this process, like ranks under one launcher); one child 'failed' with 17: all three see 17 and the SAME next port, then 0."""
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(r), "3", "17" if r == 1 else "0"], stdout=subprocess.PIPE, text=True)
             for r in range(3)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert all(o[0][0] == 17 for o in outs) and len({o[0][1] for o in outs}) == 1 and outs[0][0][1] > 0
    assert all(o[1][0] == 0 for o in outs) and len({o[1][1] for o in outs}) == 1


def test_a_missing_supervisor_ends_the_ladder():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.agree_on_outcome(0, 2, 7, 0, timeout_s=1.0) == (None, None)


def test_agreement_is_per_node():
    """the files are node-local, so the set a supervisor waits for is its node's ranks (LOCAL_RANK / LOCAL_WORLD_SIZE), never the
    global WORLD_SIZE; the directory's name carries the launcher's pid AND start time, so nothing has to be removed before use."""
    sys.path.insert(0, ROOT)
    import bench
    # a node with two local ranks of a (say) 16-rank job: the two agree without waiting for 16 files
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(r), "2", "0"], stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [json.loads(p.communicate(timeout=60)[0].strip().splitlines()[-1]) for p in procs]
    assert all(o[0][0] == 0 for o in outs)
    d = bench.sup_dir()
    assert str(os.getppid()) in os.path.basename(d) and len(os.path.basename(d).split("_")) >= 6
    assert bench.agree_on_outcome(0, 1, 99, 0, timeout_s=5.0, need_port=False) == (0, None)


def test_a_job_that_spans_nodes_does_not_wait_for_files_it_cannot_see():
    """with WORLD_SIZE = 4 and LOCAL_WORLD_SIZE = 2 the old agreement waited 420 s for rank files of the other node and then
    failed a successful run.  Now a supervisor of a multi-node job runs the first transport only and reports its own child at once.  (The
    child here has no GPU and exits 1: what matters is that the supervisor returns that code within seconds, without a fallback.)"""
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU: the rank child must fail fast")
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench, argparse; "
            "sys.argv = ['bench.py', '--gpus', '4', '--steps', '1', '--warmup', '0', '--no-cpu']; "
            "t0 = time.time(); rc = bench.supervise_rank(argparse.Namespace(gpus=4), os.dup(1)); print('RC', rc, round(time.time() - t0, 1))" % ROOT)
    env = dict(os.environ, WORLD_SIZE="4", LOCAL_WORLD_SIZE="2", RANK="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("RC ")]
    assert line and line[0].split()[1] != "0" and time.time() - t0 < 120, (r.stdout[-500:], r.stderr[-800:])


def test_summary_is_compact_and_complete():
    """the last key of the line, <= 1 KB, every record's numbers"""
    sys.path.insert(0, ROOT)
    import bench
    rec = lambda ms: {"ms_per_step": ms, "value": 1e10, "verified_against_oracle": True, "verified_rows": 44,   # noqa: E731
                      "roofline": {"frac": 0.512345678, "frac_vs_gather_ceiling": 0.7512345}}
    out = dict(rec(0.0757), config={"num_v": 169343}, no_reorder={"avg_launch_us": 85.123456},
               configs={"A_rows": rec(0.113), "R": rec(15.28), "G": rec(7.65),
                        "P1": dict(rec(7.29), no_reorder={"ms_per_step": 8.14, "frac": 0.9, "frac_vs_gather_ceiling": 1.2, "verified_rows": 204,
                                                           "verified_against_oracle": True})},
               cpu_baseline={"value": 1.5e8})
    s = bench.summarize(out)
    assert len(json.dumps(s)) <= 1024 and s["verified"] is True and s["R"] == [15.28, 0.5123, 0.7512, 44] and s["P1_no_reorder"][0] == 8.14
    out["configs"]["G"] = {"error": "boom"}
    assert bench.summarize(out)["verified"] is False


def test_byte_models_of_survey_8d():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes(169343, 1166243, 128) == 693827352      # config A
    assert bench.compulsory_bytes(169343, 1166243, 128) == 183414552


def test_float64_yardstick_matches_a_dense_product():
    sys.path.insert(0, ROOT)
    import bench
    bench.np = np
    rng = np.random.default_rng(3)
    V, F, H = 50, 8, 2
    deg = rng.integers(0, 9, 6)
    sp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    si = rng.integers(0, V, sp[-1]).astype(np.int32)
    x = rng.standard_normal((V, F)).astype(np.float32)
    v = rng.standard_normal(sp[-1]).astype(np.float32)
    w = rng.random((sp[-1], H))
    for k in range(6):
        e = slice(int(sp[k]), int(sp[k + 1]))
        assert np.allclose(bench.sum_rows_f64(sp, si, x, v)[k], (x[si[e]].astype(np.float64) * v[e, None]).sum(0))
        assert np.allclose(bench.sum_rows_f64(sp, si, x)[k], x[si[e]].astype(np.float64).sum(0))
        ref = (x[si[e]].astype(np.float64).reshape(-1, H, F // H) * w[e][:, :, None]).sum(0).reshape(F)
        assert np.allclose(bench.sum_rows_f64(sp, si, x, w=w, heads=H)[k], ref)


def test_isa_lint_for_the_cross_block_store_hazard():
    """Second tier (GNNAGG_TEST_TIER=2; ~3 minutes of hipcc): `make -C gnn_computing_amd/csrc lint` -- no kernel file's assembly has a
    store of more than 8 bytes followed across a basic-block boundary by a VALU write of its data registers, and the hand-scheduled GEMM
    kernels keep v192 .. v255 to their assembly and never touch scratch."""
    import pytest
    if os.environ.get("GNNAGG_TEST_TIER") != "2":
        pytest.skip("second tier: set GNNAGG_TEST_TIER=2 (make -C gnn_computing_amd/csrc lint)")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gnn_computing_amd", "csrc"), "lint"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "0 suspicious place(s)" in r.stdout and "0 violation(s)" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
