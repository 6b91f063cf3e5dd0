"""CPU: the product's host graph preparation (libgnnagg.so Section C/D, pure host code) against the
oracle and the golden vectors -- bit-exact (integer work)."""
import json
import os

import numpy as np
import pytest

import gnn_computing_amd as gnc
from gnn_computing_amd import _lib, graph
from oracle import oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S = json.load(open(os.path.join(G, "survey_8c.json")))
PTR, IDX = np.array(S["graph"]["ptr"], np.int32), np.array(S["graph"]["idx"], np.int32)


def test_golden_vectors_through_cabi():
    ps, tg = gnc.neighbor_grouping_schedule(PTR, 2)
    assert ps.tolist() == S["neighbor_grouping_ng2"]["ptr_s"] and tg.tolist() == S["neighbor_grouping_ng2"]["target"]
    ps, ix, tg, _ = gnc.locality_schedule(PTR, IDX, 2, 4)
    g = S["locality_par2_total4"]
    assert (ps.tolist(), ix.tolist(), tg.tolist()) == (g["ptr_s"], g["idx_s"], g["target"])
    ps, ix, tg, _ = gnc.locality_schedule(PTR, IDX, 2, 4, ng=2)
    g = S["locality_ng_par2_ng2"]
    assert (ps.tolist(), ix.tolist(), tg.tolist()) == (g["ptr_s"], g["idx_s"], g["target"])
    g = S["reorder_2031"]
    nptr, nidx, rev = gnc.reorder_csr(PTR, IDX, g["rows"])
    assert (nptr.tolist(), nidx.tolist(), rev.tolist()) == (g["ptr"], g["idx"], g["reverse_rows"])


@pytest.mark.parametrize("V,E", [(1, 0), (1, 5), (17, 0), (50, 700), (1000, 20000), (3000, 9000)])
def test_schedules_match_oracle(V, E):
    ptr, idx = graph.uniform_random_csr(V, E, seed=V + E)
    val = np.random.default_rng(1).standard_normal(E, dtype=np.float32)
    for ng in (1, 2, 7, 16, 32, 10 ** 6):
        a, b = gnc.neighbor_grouping_schedule(ptr, ng), orc.neighbor_grouping(ptr, ng)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for par, ng, total in ((1, 0, V), (2, 0, V), (3, 4, V), (7, 1, V), (5, 0, V + 13), (4, 2, max(V - 3, 1)), (V + 5, 0, V)):
        a = gnc.locality_schedule(ptr, idx, par, total, ng, val)
        b = orc.locality_schedule(ptr, idx, par, total, ng, val)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
    rows = np.random.default_rng(2).permutation(V).astype(np.int32)
    a, b = gnc.reorder_csr(ptr, idx, rows), orc.reorder_csr(ptr, idx, rows)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[3])


def test_powerlaw_arxiv_shape_schedule_matches_oracle():
    ptr, idx = graph.dataset("arxiv")
    ptr, idx = ptr.numpy(), idx.numpy()
    for ng in (16, 32):
        a, b = gnc.neighbor_grouping_schedule(ptr, ng), orc.neighbor_grouping(ptr, ng)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    rows = graph.locality_order(ptr, idx)
    assert sorted(rows.tolist()) == list(range(len(ptr) - 1))
    a, b = gnc.reorder_csr(ptr, idx, rows), orc.reorder_csr(ptr, idx, rows)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("fmt", ["text", "dumps", "both"])
def test_load_graph(tmp_path, fmt):
    d = str(tmp_path) + "/"
    ptr, idx = graph.uniform_random_csr(300, 4000, seed=3)
    graph.write_graph_files(d, "g", ptr, idx, text=fmt in ("text", "both"), dumps=fmt in ("dumps", "both"))
    rows = np.random.default_rng(4).permutation(300).astype(np.int32)
    graph.write_reorder_file(d, "g", rows)
    out = gnc.load_graph_host("g", "", d)
    assert out["num_v"] == 300 and out["num_e"] == 4000
    assert np.array_equal(out["ptr"], ptr) and np.array_equal(out["idx"], idx) and out["rows"] is None
    # caches exist afterwards with the raw int32 sizes (data.cu:64-67,:88-91)
    assert os.path.getsize(d + "g.graph.ptrdump") == 301 * 4 and os.path.getsize(d + "g.graph.edgedump") == 4000 * 4
    out = gnc.load_graph_host("g", "_thres_0.2", d)
    ref = orc.load_graph(d, "g", "_thres_0.2")
    for k in ("ptr", "idx", "rows", "reverse_rows"):
        assert np.array_equal(out[k], ref[k])
    # shuffle=False ignores the reorder file (data.cu:98)
    out = gnc.load_graph_host("g", "_thres_0.2", d, shuffle=False)
    assert np.array_equal(out["idx"], idx)
    # a suffix whose file does not exist loads un-reordered (data.cu:134-138)
    out = gnc.load_graph_host("g", "_nope", d)
    assert np.array_equal(out["idx"], idx) and out["rows"] is None


def test_load_graph_errors(tmp_path):
    d = str(tmp_path) + "/"
    with pytest.raises(gnc.GnnAggError) as ei:
        gnc.load_graph_host("missing", "", d)
    assert ei.value.code == _lib.ERR_IO
    ptr, idx = graph.uniform_random_csr(10, 30, seed=1)
    graph.write_graph_files(d, "bad", ptr, idx)
    open(d + "bad.config", "w").write("10 31")  # indptr[num_v] != num_e (data.cu:69-74)
    with pytest.raises(gnc.GnnAggError):
        gnc.load_graph_host("bad", "", d)
    graph.write_graph_files(d, "perm", ptr, idx)
    open(d + "perm.reorder_x", "w").write(" ".join(["0"] * 10))  # not a permutation
    with pytest.raises(gnc.GnnAggError):
        gnc.load_graph_host("perm", "_x", d)
    with pytest.raises(gnc.GnnAggError):
        gnc.neighbor_grouping_schedule(ptr, 0)


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
def test_partition_and_halo_plan(nparts):
    V, E = 500, 9000
    ptr, idx = graph.uniform_random_csr(V, E, seed=9)
    b = gnc.partition_rows(ptr, nparts)
    assert b[0] == 0 and b[-1] == V and np.all(np.diff(b) >= 0)
    nnz = np.diff(ptr[b])
    assert nnz.max() <= E / nparts + orc.degrees(ptr).max() + 1  # balanced up to one row
    seen_rows = 0
    for r in range(nparts):
        p = gnc.halo_plan(ptr, idx, b, r)
        n_loc = p["n_local"]
        seen_rows += n_loc
        assert np.array_equal(p["local_ptr"], ptr[b[r]:b[r + 1] + 1] - ptr[b[r]])
        glob = idx[ptr[b[r]]:ptr[b[r + 1]]]
        # translate local slots back to global ids
        slot2glob = np.concatenate([np.arange(b[r], b[r + 1]), p["halo_ids"]]).astype(np.int32)
        assert np.array_equal(slot2glob[p["local_idx"]], glob)
        assert np.all(np.diff(p["halo_ids"]) > 0)  # ascending, deduplicated
        assert not np.any((p["halo_ids"] >= b[r]) & (p["halo_ids"] < b[r + 1]))
        owners = np.searchsorted(b, p["halo_ids"], side="right") - 1
        assert np.array_equal(np.bincount(owners, minlength=nparts), p["halo_counts"])
    assert seen_rows == V


@pytest.mark.parametrize("world,mode,k", [(4, 0, 3), (8, 0, 4), (5, 1, 1), (8, 1, 1), (3, 0, 1), (1, 0, 2)])
def test_halo_stage_plan_both_ends_of_every_pair_agree(world, mode, k):
    """gnnagg_halo_stage_plan (the staged halo exchange's plan, host C++): for every pair the sender's share of a stage is the
    receiver's, every row travels in exactly one stage, the slot / send-order maps are permutations that keep a list's own order inside
    a (stage, peer) cell; stripes give every stage a slice of EVERY peer's list, owner stages name one ring distance each."""
    import ctypes
    L = gnc.lib()
    rng = np.random.default_rng(world * 10 + mode)
    rows = rng.integers(0, 50, (world, world)).astype(np.int64)     # rows[reader][owner]
    np.fill_diagonal(rows, 0)
    ns = ctypes.c_int(0)
    plans = []
    for r in range(world):
        recv, send = np.ascontiguousarray(rows[r]), np.ascontiguousarray(rows[:, r])
        assert L.gnnagg_halo_stage_plan(None, None, world, r, mode, k, ctypes.byref(ns), None, None, None, None) == 0
        S = ns.value
        assert S == (1 if world == 1 else (world - 1 if mode == 1 else k))
        st_r, st_s = np.zeros((S, world), np.int64), np.zeros((S, world), np.int64)
        perm, order = np.zeros(max(int(recv.sum()), 1), np.int32), np.zeros(max(int(send.sum()), 1), np.int32)
        assert L.gnnagg_halo_stage_plan(recv.ctypes.data, send.ctypes.data, world, r, mode, k, ctypes.byref(ns), st_r.ctypes.data, perm.ctypes.data,
                                        st_s.ctypes.data, order.ctypes.data) == 0
        assert np.array_equal(st_r.sum(axis=0), recv) and np.array_equal(st_s.sum(axis=0), send)
        n_r, n_s = int(recv.sum()), int(send.sum())
        assert sorted(perm[:n_r].tolist()) == list(range(n_r)) and sorted(order[:n_s].tolist()) == list(range(n_s))
        # inside one owner's list the new slots ascend (the rows of a list arrive in the list's own order)
        o0 = np.concatenate([[0], np.cumsum(recv)])
        assert all(np.all(np.diff(perm[o0[o]:o0[o + 1]]) > 0) for o in range(world))
        if mode == 1 and world > 1:
            assert all(np.count_nonzero(st_r[s]) <= 1 and np.count_nonzero(st_s[s]) <= 1 for s in range(S))
        plans.append((st_r, st_s))
    for r in range(world):
        for q in range(world):
            assert np.array_equal(plans[r][1][:, q], plans[q][0][:, r])      # what r sends q in a stage is what q expects from r
    assert L.gnnagg_halo_stage_plan(None, None, 0, 0, 0, 1, ctypes.byref(ns), None, None, None, None) != 0
    assert L.gnnagg_halo_stage_plan(None, None, 2, 0, 0, 0, ctypes.byref(ns), None, None, None, None) != 0


def test_reorder_on_load_goes_through_the_reference_file_formats(tmp_path):
    """gnc.graph.reorder_on_load (what bench.py's arms with the locality reorder use): the permutation of the library's generator is written
    as <dset>.reorder_thres_0.2 beside the .ptrdump / .edgedump caches and applied by gnnagg_load_graph exactly like src/data.cu:96-133 --
    arrays equal to reorder_csr on the same permutation and to the oracle's restatement of reorderCSR; a second call finds the file."""
    import gnn_computing_amd as gnc
    from oracle import oracle as orc
    ptr_t, idx_t = gnc.graph.powerlaw_csr(3000, 40000, seed=9)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    p1, i1, rows, t_gen, _ = gnc.graph.reorder_on_load("tiny", ptr, idx, key="k", cache_dir=str(tmp_path))
    assert t_gen > 0 and sorted(rows.tolist()) == list(range(3000))
    files = set(os.listdir(tmp_path))
    assert {"tiny_k.config", "tiny_k.graph.ptrdump", "tiny_k.graph.edgedump", "tiny_k.reorder_thres_0.2"} <= files
    assert np.array_equal(np.array(open(tmp_path / "tiny_k.reorder_thres_0.2").read().split(), np.int32), rows)
    q, j, _ = gnc.reorder_csr(ptr, idx, rows)
    assert np.array_equal(p1, q) and np.array_equal(i1, j)
    op, oi, _, _ = orc.reorder_csr(ptr, idx, rows)
    assert np.array_equal(p1, op) and np.array_equal(i1, oi)
    p2, i2, rows2, t_gen2, _ = gnc.graph.reorder_on_load("tiny", ptr, idx, key="k", cache_dir=str(tmp_path))
    assert t_gen2 == 0.0 and np.array_equal(rows2, rows) and np.array_equal(i2, i1)
