"""CPU: the locality reorder generator (gnnagg_cluster_reorder; reference script/cluster2.py).
datasketch's hashing is not reproducible here (parity unpinned), so the algorithmic contract is
what is pinned: a valid permutation, planted clusters come out contiguous, the cluster cap holds, the file it
produces is consumed by the loader, and aggregation results are permutation-invariant."""
import numpy as np
import pytest

import gnn_computing_amd as gnc
from gnn_computing_amd import graph
from oracle import oracle as orc


def planted(n_clusters=40, size=20, pool=8, V_extra=100, seed=0):
    rng = np.random.default_rng(seed)
    V = n_clusters * size + V_extra
    label = np.full(V, -1)
    nodes = rng.permutation(V)[:n_clusters * size].reshape(n_clusters, size)
    deg = np.zeros(V, np.int64)
    nbrs = [None] * V
    for c in range(n_clusters):
        common = rng.integers(0, V, pool)
        for v in nodes[c]:
            label[v] = c
            nbrs[v] = np.concatenate([common, rng.integers(0, V, 1)])
    for v in range(V):
        if nbrs[v] is None:
            nbrs[v] = rng.integers(0, V, rng.integers(0, 4))
        deg[v] = len(nbrs[v])
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    return ptr, np.concatenate(nbrs).astype(np.int32), label


def test_planted_clusters_become_contiguous():
    ptr, idx, label = planted()
    rows, nc = gnc.cluster_reorder(ptr, idx, cluster_cap=64)
    V = len(ptr) - 1
    assert sorted(rows.tolist()) == list(range(V))
    pos = np.empty(V, np.int64)
    pos[rows] = np.arange(V)
    spans = []
    for c in range(label.max() + 1):
        p = np.sort(pos[label == c])
        spans.append(p[-1] - p[0] + 1)
        # all 20 members sit inside ONE output cluster (<= 2*cap - 2 nodes, written contiguously); stray
        # low-similarity nodes merged into the same cluster may interleave by node id
        assert spans[-1] <= 2 * 64 - 2, "cluster %d is scattered after the reorder (span %d)" % (c, spans[-1])
    # (clusters below the cap keep absorbing their most similar neighbours, so several planted groups usually
    # share one output cluster and interleave inside it -- same as the reference script)
    assert nc < V  # something was merged


def test_cluster_cap():
    ptr, idx, label = planted(n_clusters=5, size=100, pool=10, V_extra=0)
    rows, nc = gnc.cluster_reorder(ptr, idx, cluster_cap=8)
    # merging stops once a cluster reaches the cap: a merge of two clusters below the cap gives at most 2*cap - 2
    assert nc >= (len(ptr) - 1) / (2 * 8 - 2)


def test_reorder_file_roundtrip_and_invariance(tmp_path):
    ptr, idx = graph.powerlaw_csr(3000, 30000, seed=4)
    ptr, idx = ptr.numpy(), idx.numpy()
    rows, _ = gnc.cluster_reorder(ptr, idx)
    d = str(tmp_path) + "/"
    graph.write_graph_files(d, "g", ptr, idx)
    graph.write_reorder_file(d, "g", rows)  # <dset>.reorder_thres_0.2, the name the reference loads (our.py:79)
    out = gnc.load_graph_host("g", "_thres_0.2", d)
    nptr, nidx, rev = gnc.reorder_csr(ptr, idx, rows)
    assert np.array_equal(out["ptr"], nptr) and np.array_equal(out["idx"], nidx) and np.array_equal(out["rows"], rows)
    # aggregation commutes with the relabelling (validReordered's contract, spmm.h:23-33)
    x = np.random.default_rng(1).standard_normal((3000, 8), dtype=np.float32)
    y, y2 = orc.gcn_seq(ptr, idx, None, x), orc.gcn_seq(nptr, nidx, None, x[rows])
    assert np.array_equal(y2, y[rows])
    assert orc.validate_reordered(y, y2, rev) == 0


def test_degenerate_inputs():
    rows, nc = gnc.cluster_reorder(np.zeros(6, np.int32), np.zeros(0, np.int32))
    assert rows.tolist() == [0, 1, 2, 3, 4] and nc == 5  # empty rows are never queried (cluster2.py:83-84)
    rows, nc = gnc.cluster_reorder(np.array([0], np.int32), np.zeros(0, np.int32))
    assert len(rows) == 0


def _lru_hit_rate(ptr, idx, cap):
    """Share of the gathers that find their source row among the `cap` most recently gathered ones (rows in order)."""
    from collections import OrderedDict
    lru, hits = OrderedDict(), 0
    for s in idx.tolist():
        if s in lru:
            hits += 1
            lru.move_to_end(s)
        else:
            lru[s] = 1
            if len(lru) > cap:
                lru.popitem(last=False)
    return hits / max(len(idx), 1)


@pytest.mark.parametrize("cluster_cap", [1, 64])
def test_cache_greedy_order_is_a_permutation_and_finds_hidden_locality(cluster_cap):
    """order_mode 1 (gnnagg_cluster_reorder_ex): a hidden ring order with windowed neighbor sets, scattered by a random
    relabelling.  The cache-aware greedy order must be a valid permutation and recover a large part of the reuse the hidden
    order has -- far more than the first-member order of the same clusters."""
    V, deg, window = 6000, 8, 64
    rng = np.random.default_rng(5)
    sigma = rng.permutation(V)                      # node -> hidden position
    inv = np.argsort(sigma)
    ptr = np.arange(0, (V + 1) * deg, deg, dtype=np.int32)
    pos = (sigma[:, None] + rng.integers(-window, window + 1, (V, deg))) % V
    idx = np.sort(inv[pos], axis=1).astype(np.int32).ravel()
    cap_rows = 256
    rows_g, nc_g = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=cluster_cap, cache_rows=cap_rows)
    rows_f, nc_f = gnc.cluster_reorder(ptr, idx, order="first_member", cluster_cap=cluster_cap)
    assert sorted(rows_g.tolist()) == list(range(V)) and nc_g == nc_f
    if cluster_cap == 1:
        assert nc_g == V and np.array_equal(rows_f, np.arange(V))      # singleton clusters: first-member order = identity
    hit = {}
    for name, rows in (("greedy", rows_g), ("first", rows_f), ("hidden", inv.astype(np.int32))):
        p, i, _ = gnc.reorder_csr(ptr, idx, rows)
        hit[name] = _lru_hit_rate(p, i, cap_rows)
    assert hit["hidden"] > 0.8
    assert hit["greedy"] > 0.6 * hit["hidden"] and hit["greedy"] > hit["first"] + 0.05
    if cluster_cap == 1:
        assert hit["first"] < 0.3 and hit["greedy"] > 2 * hit["first"]   # (the scattered numbering itself has no reuse)


def test_cluster_reorder_ex_argument_checks():
    import ctypes
    from gnn_computing_amd._lib import check
    ptr, idx, _ = planted(4, 6, 10, 3)
    rows = np.empty(len(ptr) - 1, np.int32)
    with pytest.raises(gnc.GnnAggError):   # unknown order mode
        check(gnc.lib().gnnagg_cluster_reorder_ex(ptr.ctypes.data, idx.ctypes.data, len(ptr) - 1, ctypes.c_float(0.2), 64, 64,
                                                  ctypes.c_ulonglong(1), 7, 4096, rows.ctypes.data, None))


def _window_footprint(ptr, idx, order, W=4096, long_row=64):
    """distinct sources / edges over windows of W consecutive rows of `order` (rows beyond long_row edges left out, as in the
    generator's cache model): what a window of rows in flight asks of the cache -- lower is better."""
    deg = np.diff(ptr)
    d = e = 0
    for w0 in range(0, len(order), W):
        rows = order[w0:w0 + W]
        rows = rows[deg[rows] <= long_row]
        if len(rows):
            src = np.concatenate([idx[ptr[r]:ptr[r + 1]] for r in rows])
            d += len(np.unique(src))
            e += len(src)
    return d / max(e, 1)


def test_parallel_walkers_keep_the_quality_of_the_serial_greedy_order(monkeypatch):
    """order_mode 1 with several walkers (reorder.cpp, emit_cache_greedy_parallel: the products-shaped graph went from 5.4 minutes
    of one core to about a minute on 8): a valid permutation whose window footprint stays with the serial pass's -- far below the
    scattered input order's.  The walkers are logical and advance in bulk-synchronous rounds, so the order is a function of the
    input and the walker count alone: the same with one thread, three or all of them."""
    import ctypes
    V, E = 60000, 1800000
    ptr_t, idx_t = graph.powerlaw_csr(V, E, seed=11)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    monkeypatch.setenv("GNNAGG_REORDER_WALKERS", "1")
    serial, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=4096)
    serial2, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=4096)
    assert np.array_equal(serial, serial2)                                   # the single-walker path is deterministic
    monkeypatch.setenv("GNNAGG_REORDER_WALKERS", "4")
    par, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=4096)
    assert np.array_equal(np.sort(par), np.arange(V)) and np.array_equal(np.sort(serial), np.arange(V))
    p64, i64 = ptr.astype(np.int64), idx.astype(np.int64)
    f_id, f_ser, f_par = (_window_footprint(p64, i64, o.astype(np.int64)) for o in (np.arange(V), serial, par))
    assert f_ser < 0.8 * f_id and f_par < f_ser * 1.06, (f_id, f_ser, f_par)
    gomp = ctypes.CDLL("libgomp.so.1")                                       # the OpenMP runtime the library is linked against
    gomp.omp_get_max_threads.restype = ctypes.c_int
    n0 = gomp.omp_get_max_threads()
    try:
        for threads in (1, 3):
            gomp.omp_set_num_threads(threads)
            again, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=4096)
            assert np.array_equal(again, par), "4 walkers on %d thread(s) gave another order" % threads
    finally:
        gomp.omp_set_num_threads(n0)
