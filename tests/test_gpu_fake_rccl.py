"""The librccl TEST DOUBLE checked by itself (tests/fake_rccl/selftest.cpp), before tests/test_gpu_dist.py relies on it: ranks are
processes sharing the one GPU; a message is a stream-ordered device-to-device copy out of the sender's IPC-mapped buffer, ordered
between the two processes' streams by counters in IPC-shared device memory -- no host-side synchronisation inside a group (VERDICT r5
item 1).  The self-test proves the three properties the step tests lean on: (1) back-to-back rounds with nothing but stream order
between them deliver the right words, (2) ncclGroupEnd returns while a late peer's data is NOT there yet (the poison is still in the
receive buffer), (3) a captured round replays."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "fake_rccl", "selftest.out")


def run_selftest(world, tmp_path, rounds=20, extra_env=None, timeout=300):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    idfile = str(tmp_path / "id")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), **(extra_env or {}))
        procs.append(subprocess.Popen([EXE, "--idfile", idfile, "--rounds", str(rounds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    out = []
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=timeout)
        lines = [json.loads(l) for l in so.splitlines() if l.startswith("{")]
        assert p.returncode == 0 and lines, "rank %d: exit %s\n%s\n%s" % (r, p.returncode, so[-2000:], se[-2000:])
        out.append(lines[-1])
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_double_is_stream_ordered_and_asynchronous(world, tmp_path):
    res = run_selftest(world, tmp_path)
    for r in res:
        assert r["ok"] and r["bad_words_rounds"] == 0 and r["bad_words_async"] == 0 and r["bad_words_graph"] == 0 and r["graph_replays"] == 5
        if r["rank"] != 1:
            # the late peer's words were still poison when ncclGroupEnd had returned: the host did not wait for them
            assert r["poison_words_expected"] > 0 and r["poison_words_seen_after_group_end"] == r["poison_words_expected"]
            assert r["group_end_ms_with_late_peer"] < 150.0


def test_double_with_every_rank_late(tmp_path):
    """every group of every rank behind a 2 ms spin kernel (FAKE_RCCL_DELAY_US): the rounds still deliver the right words"""
    res = run_selftest(3, tmp_path, rounds=10, extra_env={"FAKE_RCCL_DELAY_US": "2000"})
    assert all(r["ok"] for r in res)
