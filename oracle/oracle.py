"""ctypes/numpy front-end of the CPU oracle (oracle/gnn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (gnn_computing_amd) never imports this module.

The file-format half of the loader (reference src/data.cu:31-93) is restated here in numpy
because it is byte/integer parsing; everything numeric lives in the C file.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

_I = ctypes.POINTER(ctypes.c_int)
_F = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "gnn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_neighbor_grouping.restype = ctypes.c_int
        _lib.orc_locality_schedule.restype = ctypes.c_int
        _lib.orc_validate2.restype = ctypes.c_int
        _lib.orc_validate_reordered.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _i(a):
    return None if a is None else a.ctypes.data_as(_I)


def _f(a):
    return None if a is None else a.ctypes.data_as(_F)


def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _cf(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---------------------------------------------------------------- integer stages
def reverse_map(rows):
    rows = _ci(rows)
    out = np.empty_like(rows)
    lib().orc_reverse_map(_i(rows), ctypes.c_int(len(rows)), _i(out))
    return out


def reorder_csr(ptr, idx, rows):
    """reference src/data.cu:4-29 + :105-113.  Returns (newptr, newidx, rows, reverse_rows)."""
    ptr, idx, rows = _ci(ptr), _ci(idx), _ci(rows)
    V, E = len(ptr) - 1, len(idx)
    rev = reverse_map(rows)
    newptr = np.empty(V + 1, np.int32)
    newidx = np.empty(E, np.int32)
    lib().orc_reorder_csr(_i(ptr), _i(idx), _i(rows), _i(rev), V, E, _i(newptr), _i(newidx))
    return newptr, newidx, rows, rev


def neighbor_grouping(ptr, ng):
    """reference include/graph_schedule.h:91-126.  Returns (ptr_s[G+1], target[G])."""
    ptr = _ci(ptr)
    V = len(ptr) - 1
    G = lib().orc_neighbor_grouping(_i(ptr), int(ng), V, None, None)
    ptr_s = np.empty(G + 1, np.int32)
    tgt = np.empty(G, np.int32)
    lib().orc_neighbor_grouping(_i(ptr), int(ng), V, _i(ptr_s), _i(tgt))
    return ptr_s, tgt


def locality_schedule(ptr, idx, par_num, total_v, ng=0, val=None):
    """reference include/graph_schedule.h:17-63 (ng=0) / :156-211 (ng>0).
    Returns (ptr_s, idx_s, target, val_s or None)."""
    ptr, idx, val = _ci(ptr), _ci(idx), _cf(val)
    V, E = len(ptr) - 1, len(idx)
    ptr_s = np.empty(E + 2, np.int32)
    idx_s = np.empty(max(E, 1), np.int32)
    tgt = np.empty(max(E, 1), np.int32)
    val_s = np.empty(max(E, 1), np.float32) if val is not None else None
    G = lib().orc_locality_schedule(_i(ptr), _i(idx), _f(val), int(par_num), int(ng), V, int(total_v),
                                    _i(ptr_s), _i(idx_s), _f(val_s), _i(tgt))
    n = int(ptr_s[G])
    return ptr_s[:G + 1].copy(), idx_s[:n].copy(), tgt[:G].copy(), (None if val is None else val_s[:n].copy())


def csr2edgelist(ptr, idx):
    ptr, idx = _ci(ptr), _ci(idx)
    out = np.empty(2 * len(idx), np.int32)
    lib().orc_csr2edgelist(_i(ptr), _i(idx), len(ptr) - 1, _i(out))
    return out


def degrees(ptr):
    ptr = _ci(ptr)
    out = np.empty(len(ptr) - 1, np.int32)
    lib().orc_degrees(_i(ptr), len(ptr) - 1, _i(out))
    return out


# ---------------------------------------------------------------- GCN
def _gcn(fn, ptr, idx, val, X):
    ptr, idx, val, X = _ci(ptr), _ci(idx), _cf(val), _cf(X)
    V, F = len(ptr) - 1, X.shape[1]
    Y = np.empty((V, F), np.float32)
    fn(_i(ptr), _i(idx), _f(val), _f(X), _f(Y), V, F)
    return Y


def gcn_seq(ptr, idx, val, X):
    return _gcn(lib().orc_gcn_seq, ptr, idx, val, X)


def gcn_mean(ptr, idx, val, X):
    return _gcn(lib().orc_gcn_mean, ptr, idx, val, X)


def gcn_max(ptr, idx, val, X):
    return _gcn(lib().orc_gcn_max, ptr, idx, val, X)


def gcn_abs_scale(ptr, idx, val, X):
    return _gcn(lib().orc_gcn_abs_scale, ptr, idx, val, X)


def gcn_grouped(ptr_s, target, idx, val, X, num_v, seg=0):
    """seg > 0: partials folded inside segments of `seg` groups per row, then the segment sums (the balanced
    mode's order, see orc_gcn_grouped_seg); seg = 0: flat ascending fold (reference NG semantics)."""
    ptr_s, target, idx, val, X = _ci(ptr_s), _ci(target), _ci(idx), _cf(val), _cf(X)
    F = X.shape[1]
    Y = np.empty((num_v, F), np.float32)
    lib().orc_gcn_grouped_seg(_i(ptr_s), _i(target), len(target), _i(idx), _f(val), _f(X), _f(Y), int(num_v), F,
                              int(seg))
    return Y


def matmul_nn(A, B):
    A, B = _cf(A), _cf(B)
    C = np.empty((A.shape[0], B.shape[1]), np.float32)
    lib().orc_matmul_nn(_f(A), _f(B), _f(C), A.shape[0], B.shape[1], A.shape[1])
    return C


def spmm_naive(ptr, idx, val, X, Y_init):
    ptr, idx, val, X = _ci(ptr), _ci(idx), _cf(val), _cf(X)
    Y = np.array(Y_init, dtype=np.float32, order="C", copy=True)
    lib().orc_spmm_naive(_i(ptr), _i(idx), _f(val), _f(X), _f(Y), len(ptr) - 1, X.shape[1])
    return Y


def validate2(ref, ans):
    ref, ans = _cf(ref).ravel(), _cf(ans).ravel()
    with np.errstate(all="ignore"):
        return lib().orc_validate2(_f(ref), _f(ans), len(ref))


def validate_reordered(ref, ans, rows):
    ref, ans, rows = _cf(ref), _cf(ans), _ci(rows)
    return lib().orc_validate_reordered(_f(ref), _f(ans), _i(rows), ref.shape[0], ref.shape[1])


# ---------------------------------------------------------------- GAT
def gat_fused(ptr, idx, att, X, heads=1, slope=0.2):
    ptr, idx, att, X = _ci(ptr), _ci(idx), _cf(att), _cf(X)
    V, F = len(ptr) - 1, X.shape[1]
    Y = np.empty((V, F), np.float32)
    lib().orc_gat_fused(_i(ptr), _i(idx), _f(att), _f(X), _f(Y), V, int(heads), F // heads, ctypes.c_float(slope))
    return Y


def gat_att(ptr, idx, att, heads=1, slope=0.2):
    ptr, idx, att = _ci(ptr), _ci(idx), _cf(att)
    out = np.empty((len(idx), heads), np.float32)
    lib().orc_gat_att(_i(ptr), _i(idx), _f(att), _f(out), len(ptr) - 1, int(heads), ctypes.c_float(slope))
    return out


def gat_u_add_v(ptr, idx, att):
    ptr, idx, att = _ci(ptr), _ci(idx), _cf(att)
    out = np.empty(len(idx), np.float32)
    lib().orc_gat_u_add_v(_i(ptr), _i(idx), _f(att), _f(out), len(ptr) - 1)
    return out


def gat_add_to_center(ptr, newval):
    ptr, newval = _ci(ptr), _cf(newval)
    out = np.empty(len(ptr) - 1, np.float32)
    lib().orc_gat_add_to_center(_i(ptr), _f(newval), _f(out), len(ptr) - 1)
    return out


def gat_div_each(ptr, center, newval):
    ptr, center = _ci(ptr), _cf(center)
    out = np.array(newval, dtype=np.float32, copy=True)
    with np.errstate(all="ignore"):
        lib().orc_gat_div_each(_i(ptr), _f(center), _f(out), len(ptr) - 1)
    return out


def gat_bwd(ptr, idx, output, doutput, newval, div, infeat, slope=0.2):
    """Backward of the single-head fused GAT aggregation (orc_gat_bwd).  Returns (d_a_b[V,2], d_feat[V,F])."""
    ptr, idx = _ci(ptr), _ci(idx)
    output, doutput, newval, div, infeat = _cf(output), _cf(doutput), _cf(newval).ravel(), _cf(div).ravel(), _cf(infeat)
    V, F = len(ptr) - 1, infeat.shape[1]
    d_a_b = np.empty((V, 2), np.float32)
    d_feat = np.empty((V, F), np.float32)
    lib().orc_gat_bwd(_i(ptr), _i(idx), _f(output), _f(doutput), _f(newval), _f(div), _f(infeat), _f(d_a_b), _f(d_feat), V, F,
                      ctypes.c_float(slope))
    return d_a_b, d_feat


def gat_grouped(ptr_s, target, idx, att, X, num_v, heads=1, slope=0.2, seg=0, parts=False):
    """Returns (Y, newval[E,H] un-normalised, scalar[V,H]); seg as in gcn_grouped.  parts=True: the third element is
    (numerator[V,F], denominator[V,H]) in float64 instead -- Y times its denominator, for checks of the two-pass form
    (tolerance-compared: the product re-rounds)."""
    ptr_s, target, idx, att, X = _ci(ptr_s), _ci(target), _ci(idx), _cf(att), _cf(X)
    F = X.shape[1]
    Y = np.empty((num_v, F), np.float32)
    newval = np.zeros((len(idx), heads), np.float32)
    scalar = np.empty((num_v, heads), np.float32)
    lib().orc_gat_grouped_seg(_i(ptr_s), _i(target), len(target), _i(idx), _f(att), _f(X), _f(Y), _f(newval),
                              _f(scalar), int(num_v), int(heads), F // heads, ctypes.c_float(slope), int(seg))
    if parts:
        den = scalar.astype(np.float64)
        return Y, newval, (Y.astype(np.float64) * np.repeat(den, F // heads, axis=1), den)
    return Y, newval, scalar


# ---------------------------------------------------------------- loader (numpy restatement)
def load_graph(datadir, dset, reorder_suffix=""):
    """reference src/data.cu:31-139.  Reads <dir><dset>.config ("V E"), then ptr/idx from the raw
    int32 caches <dset>.graph.ptrdump/.edgedump if present, else from the two-line text file
    <dset>.graph, writing the caches (:64-67,:88-91).  If <dset>.reorder<suffix> exists and a
    suffix is given, applies reorderCSR (:96-133).  Returns dict(ptr, idx, rows, reverse_rows)."""
    base = os.path.join(datadir, dset)
    with open(base + ".config") as f:
        V, E = [int(t) for t in f.read().split()[:2]]
    graph = base + ".graph"
    toks = None
    if os.path.exists(graph + ".ptrdump"):
        ptr = np.fromfile(graph + ".ptrdump", dtype="<i4", count=V + 1)
    else:
        toks = np.array(open(graph).read().split(), dtype=np.int64)
        ptr = toks[:V + 1].astype(np.int32)
        ptr.astype("<i4").tofile(graph + ".ptrdump")
    assert int(ptr[V]) == E, "indptr[num_v] != num_e (data.cu:69-74)"
    if os.path.exists(graph + ".edgedump"):
        idx = np.fromfile(graph + ".edgedump", dtype="<i4", count=E)
    else:
        if toks is None:
            toks = np.array(open(graph).read().split(), dtype=np.int64)
        idx = toks[V + 1:V + 1 + E].astype(np.int32)
        idx.astype("<i4").tofile(graph + ".edgedump")
    rows = rev = None
    rfile = base + ".reorder" + reorder_suffix
    if reorder_suffix and os.path.exists(rfile):
        rows = np.array(open(rfile).read().split(), dtype=np.int32)[:V]
        ptr, idx, rows, rev = reorder_csr(ptr, idx, rows)
    return dict(ptr=ptr, idx=idx, rows=rows, reverse_rows=rev, num_v=V, num_e=E)
