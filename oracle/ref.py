"""ctypes front-end of oracle/_ref/libref_gnn.so -- the REFERENCE's own code (hipify-perl translation of /root/reference,
built by oracle/ref_build.sh), used to PIN the CPU oracle and the product:

* host-only entry points (the reference's schedulers, reorderCSR, load_graph) run anywhere the library loads;
* device entry points (Aggregator_GCN::run, Aggregator_GAT::run, csr2edgelist) need a GPU.

TEST INFRASTRUCTURE ONLY: imported by tests/ and by tests/golden/make_reference_vectors.py.  The product package never
imports this module.  `available()` is False when the library was not built (no reference tree at build time).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libref_gnn.so")
_lib = None

_I = ctypes.POINTER(ctypes.c_int)
_F = ctypes.POINTER(ctypes.c_float)


def build():
    """Runs the committed recipe (needs /root/reference and hipify-perl; a no-op message otherwise)."""
    subprocess.check_call(["bash", os.path.join(_HERE, "ref_build.sh")])
    return os.path.exists(_SO)


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_gnn.so is not built (oracle/ref_build.sh needs the reference tree)")
        _lib = ctypes.CDLL(_SO)
        for name in ("ref_device_count", "ref_neighbor_grouping", "ref_locality_schedule", "ref_locality_neighbor_grouping",
                     "ref_reorder_csr", "ref_load_graph", "ref_gcn_run", "ref_csr2edgelist", "ref_gat_run", "ref_gat_edge_stage", "ref_time_run", "ref_spmm_naive", "ref_valid", "ref_gcn_variant", "ref_matmul_nn"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _i(a):
    return a.ctypes.data_as(_I)


def _f(a):
    return a.ctypes.data_as(_F)


def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _cf(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    return int(lib().ref_device_count())


def _groups(call, num_e, cap):
    ptr_s, idx_s, target = np.zeros(cap + 1, np.int32), np.zeros(max(num_e, 1), np.int32), np.zeros(max(cap, 1), np.int32)
    g = call(_i(ptr_s), _i(idx_s), _i(target), cap)
    if g < 0:
        raise RuntimeError("group capacity too small")
    return ptr_s[:g + 1].copy(), idx_s[:num_e].copy(), target[:g].copy()


def neighbor_grouping(ptr, idx, ng):
    """graph_schedule.h:91-126 -> (ptr_s[G+1], idx_s[E], target[G])"""
    ptr, idx = _ci(ptr), _ci(idx)
    V, E = len(ptr) - 1, len(idx)
    cap = V + E // max(ng, 1) + 8
    return _groups(lambda p, i, t, c: lib().ref_neighbor_grouping(_i(ptr), _i(idx), int(ng), V, E, p, i, t, c), E, cap)


def locality_schedule(ptr, idx, par_num, total_cols, ng=0):
    """graph_schedule.h:17-89 (ng == 0) / localityNeighborGrouping :156-243 -> (ptr_s, idx_s, target)"""
    ptr, idx = _ci(ptr), _ci(idx)
    V, E = len(ptr) - 1, len(idx)
    cap = V * par_num + (E // ng if ng else 0) + 8
    if ng:
        return _groups(lambda p, i, t, c: lib().ref_locality_neighbor_grouping(_i(ptr), _i(idx), int(par_num), int(ng), V, int(total_cols),
                                                                                p, i, t, c, E), E, cap)
    return _groups(lambda p, i, t, c: lib().ref_locality_schedule(_i(ptr), _i(idx), int(par_num), V, int(total_cols), p, i, t, c, E), E, cap)


def reorder_csr(ptr, idx, rows):
    """src/data.cu:4-29 with map = rows (old id placed at new position i) and its inverse"""
    ptr, idx, rows = _ci(ptr), _ci(idx), _ci(rows)
    V, E = len(ptr) - 1, len(idx)
    rev = np.empty(V, np.int32)
    rev[rows] = np.arange(V, dtype=np.int32)
    newptr, newidx = np.zeros(V + 1, np.int32), np.zeros(max(E, 1), np.int32)
    lib().ref_reorder_csr(_i(ptr), _i(idx), _i(rows), _i(rev), V, E, _i(newptr), _i(newidx))
    return newptr, newidx[:E].copy()


def load_graph(datadir, dset, shuffle=True, reorder_suffix="", cap_v=1 << 22, cap_e=1 << 26):
    """src/data.cu:31-139.  `datadir` must be a directory NAMED data (the reference reads "../data/<dset>.*"); returns
    (ptr, idx, rows or None, reverse_rows or None)."""
    datadir = os.path.abspath(datadir)
    assert os.path.basename(datadir) == "data", "the reference hard-codes ../data/"
    work = os.path.join(os.path.dirname(datadir), "run")
    os.makedirs(work, exist_ok=True)
    with open(os.path.join(datadir, dset + ".config")) as f:
        V, E = [int(t) for t in f.read().split()[:2]]
    cap_v, cap_e = max(V, 1), max(E, 1)
    nv, ne = ctypes.c_int(0), ctypes.c_int(0)
    ptr, idx = np.zeros(cap_v + 1, np.int32), np.zeros(cap_e, np.int32)
    rows, rev = np.zeros(cap_v, np.int32), np.zeros(cap_v, np.int32)
    rc = lib().ref_load_graph(work.encode(), dset.encode(), 1 if shuffle else 0, reorder_suffix.encode(), ctypes.byref(nv), ctypes.byref(ne),
                              _i(ptr), cap_v, _i(idx), cap_e, _i(rows), _i(rev))
    if rc < 0:
        raise RuntimeError("ref_load_graph failed (%d)" % rc)
    V, E = nv.value, ne.value
    return ptr[:V + 1].copy(), idx[:E].copy(), (rows[:V].copy() if rc == 1 else None), (rev[:V].copy() if rc == 1 else None)


def gcn_run(ptr, idx, val, x, block=512, scheduled=False, ng=16, want_schedule=False, par=0):
    """Aggregator_GCN::run (aggr_gcn.h:379-410): aggr_gcn, or schedule(neighbor_grouping, ng) + aggr_gcn_target.  feat must be a
    multiple of 32 with block % feat == 0 (the reference's launch geometry).  Needs a GPU."""
    ptr, idx, val, x = _ci(ptr), _ci(idx), _cf(val), _cf(x)
    V, E, F = len(ptr) - 1, len(idx), x.shape[1]
    assert F % 32 == 0 and block % F == 0 and block <= 1024
    y = np.zeros((V, F), np.float32)
    cap = V * max(par, 1) + E // max(ng, 1) + 8
    sp, st = np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
    kind = 0 if not scheduled else (2 if par > 0 else 1)   # par > 0: schedule(locality_neighbor_grouping, {par, ng})
    g = lib().ref_gcn_run(_i(ptr), _i(idx), _f(val), V, E, _f(x), _f(y), F, int(block), kind, int(ng), _i(sp), _i(st), cap, int(par))
    if g < 0:
        raise RuntimeError("ref_gcn_run failed (%d)" % g)
    if want_schedule:
        return y, sp[:g + 1].copy(), st[:g].copy()
    return y


def csr2edgelist(ptr, idx):
    ptr, idx = _ci(ptr), _ci(idx)
    V, E = len(ptr) - 1, len(idx)
    out = np.zeros(2 * max(E, 1), np.int32)
    if lib().ref_csr2edgelist(_i(ptr), _i(idx), V, E, _i(out)) < 0:
        raise RuntimeError("ref_csr2edgelist failed")
    return out[:2 * E].reshape(E, 2).copy()


def gat_run(ptr, idx, att, x, block=128, scheduled=False, ng=32):
    """Aggregator_GAT::run (aggr_gat.h:317-354): aggr_gat, or schedule(neighbor_grouping, ng) + aggr_gat_fine + scaleArray.
    att is [V, 2].  Needs a GPU."""
    ptr, idx, att, x = _ci(ptr), _ci(idx), _cf(att), _cf(x)
    V, E, F = len(ptr) - 1, len(idx), x.shape[1]
    assert F % 32 == 0 and block % F == 0 and att.shape == (V, 2)
    y = np.zeros((V, F), np.float32)
    g = lib().ref_gat_run(_i(ptr), _i(idx), V, E, _f(x), _f(att), _f(y), F, int(block), 1 if scheduled else 0, int(ng))
    if g < 0:
        raise RuntimeError("ref_gat_run failed (%d)" % g)
    return y


def _edge_stage(what, ptr, idx, vec, val):
    ptr, idx = _ci(ptr), _ci(idx)
    V, E = len(ptr) - 1, len(idx)
    vec, val = np.array(vec, dtype=np.float32, order="C"), np.array(val, dtype=np.float32, order="C")
    if val.size == 0:
        val = np.zeros(1, np.float32)
    if lib().ref_gat_edge_stage(int(what), _i(ptr), _i(idx), V, E, _f(vec), _f(val)) < 0:
        raise RuntimeError("ref_gat_edge_stage(%d) failed" % what)
    return vec, val[:E]


def gat_att(ptr, idx, att):
    """Aggregator_GAT::run_att -> attGat (aggr_gat.h:5-31,395-401), BLOCK_SIZE 32: normalised edge weights [E]."""
    return _edge_stage(0, ptr, idx, att, np.zeros(len(idx), np.float32))[1]


def gat_u_add_v(ptr, idx, att):
    """run_u_add_v -> u_add_v (aggr_gat.h:33-48): att[row, 0] + att[src, 1] per edge."""
    return _edge_stage(1, ptr, idx, att, np.zeros(len(idx), np.float32))[1]


def gat_add_to_center(ptr, idx, val):
    """run_add_to_center -> add_to_center (aggr_gat.h:50-74): row sums of val[E] -> [V]."""
    return _edge_stage(2, ptr, idx, np.zeros(len(ptr) - 1, np.float32), val)[0]


def gat_div_each(ptr, idx, vec, val):
    """run_div_each -> each_div (aggr_gat.h:76-92): val[e] / vec[row]."""
    return _edge_stage(3, ptr, idx, vec, val)[1]


def time_run(kind, ptr, idx, aux, x, block, scheduled=False, ng=16, warm=10, iters=10):
    """Microseconds per Aggregator_GCN::run (kind "gcn", aux = val[E]) / Aggregator_GAT::run ("gat", aux = att[V,2]) on this GPU
    (the reference's protocol: 10 warm-up + 10 timed calls, Figure10/main_a.cu:73-92)."""
    ptr, idx, aux, x = _ci(ptr), _ci(idx), _cf(aux), _cf(x)
    V, E, F = len(ptr) - 1, len(idx), x.shape[1]
    assert F % 32 == 0 and block % F == 0 and block <= 1024
    us = ctypes.c_double(0.0)
    rc = lib().ref_time_run(0 if kind == "gcn" else 1, _i(ptr), _i(idx), _f(aux), V, E, _f(x), F, int(block), 1 if scheduled else 0, int(ng),
                            int(warm), int(iters), ctypes.byref(us))
    if rc < 0:
        raise RuntimeError("ref_time_run failed (%d)" % rc)
    return us.value


def spmm_naive(ptr, idx, val, x, y0):
    """spmm<L> (spmm.h:223-265), L = feat in {32, 64, 128}; rows without edges keep y0."""
    ptr, idx, val, x = _ci(ptr), _ci(idx), _cf(val), _cf(x)
    y = np.array(y0, dtype=np.float32, order="C")
    rc = lib().ref_spmm_naive(_i(ptr), _i(idx), _f(val), len(ptr) - 1, len(idx), _f(x), _f(y), x.shape[1])
    if rc < 0:
        raise RuntimeError("ref_spmm_naive failed (%d)" % rc)
    return y


def valid(ref_y, ans, rows=None):
    """valid (spmm.h:35-71) or, with the loader's rows[] map, validReordered (:73-91): the mismatch count."""
    ref_y, ans = _cf(ref_y), _cf(ans)
    rows = None if rows is None else _ci(rows)
    V, F = ref_y.shape
    n = lib().ref_valid(_f(ref_y), _f(ans), None if rows is None else _i(rows), V, F)
    if n < 0:
        raise RuntimeError("ref_valid failed (%d)" % n)
    return n


def gcn_edgewise(ptr, idx, val, x, block=512):
    """Aggregator_GCN::runEdgeWise (aggr_gcn.h:446-460): 32 columns; E must be a multiple of block / 32."""
    ptr, idx, val, x = _ci(ptr), _ci(idx), _cf(val), _cf(x)
    V, E, F = len(ptr) - 1, len(idx), x.shape[1]
    assert F == 32 and E % (block // 32) == 0
    y = np.zeros((V, F), np.float32)
    if lib().ref_gcn_variant(0, _i(ptr), _i(idx), _f(val), V, E, _f(x), _f(y), F, int(block), 0, None, None, 0) < 0:
        raise RuntimeError("ref_gcn_variant(edgewise) failed")
    return y


def gcn_run_with_nn(ptr, idx, val, x, weight, block=128, ng=64):
    """schedule(neighbor_grouping, ng) + Aggregator_GCN::run_with_nn (aggr_gcn.h:491-499) -> (vout [V, F], transformed [V, out])."""
    ptr, idx, val, x, weight = _ci(ptr), _ci(idx), _cf(val), _cf(x), _cf(weight)
    V, E, F = len(ptr) - 1, len(idx), x.shape[1]
    out = weight.shape[1]
    assert F % 32 == 0 and block % F == 0 and weight.shape[0] == F and out <= 32
    y, tr = np.zeros((V, F), np.float32), np.zeros((V, out), np.float32)
    if lib().ref_gcn_variant(1, _i(ptr), _i(idx), _f(val), V, E, _f(x), _f(y), F, int(block), int(ng), _f(weight), _f(tr), out) < 0:
        raise RuntimeError("ref_gcn_variant(run_with_nn) failed")
    return y, tr


def matmul_nn(a, b):
    """matmul_NN (dense.h:4-23): row-major C = A . B through the vendor GEMM + transposing geam."""
    a, b = _cf(a), _cf(b)
    M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == K
    c = np.zeros((M, N), np.float32)
    if lib().ref_matmul_nn(_f(a), _f(b), _f(c), M, N, K) < 0:
        raise RuntimeError("ref_matmul_nn failed")
    return c
