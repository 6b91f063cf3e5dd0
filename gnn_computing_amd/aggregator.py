"""Host-side mirror of the reference's operator interface, over the C-ABI of libgnnagg.so.

Two surfaces, both taking torch tensors that live in HIP device memory (torch is only the
allocator / stream provider here; all compute happens in the hand-written HIP kernels):

* classes ``Aggregator_GCN`` / ``Aggregator_GAT`` with the method names, argument order and
  semantics of reference include/aggr_gcn.h:362-550 and include/aggr_gat.h:299-441;
* the flat functions the reference's pybind module exports (Figure7/kernel.cpp:166-179):
  ``new_load, gcn_init, gcn_update_val, gcn_run, gcn_schedule, gat_init, gat_run, gat_schedule,
  gat_run_u_add_v, gat_run_add_to_center, gat_run_div_each`` -- a script written against the
  reference extension (Figure7/our.py:79-84,171-188) runs unchanged with
  ``import gnn_computing_amd as gnc``.
"""
import ctypes
import enum

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


class Schedule(enum.IntEnum):
    """reference include/graph_schedule.h:8-14"""
    locality = _lib.SCHED_LOCALITY
    neighbor_grouping = _lib.SCHED_NEIGHBOR_GROUPING
    locality_neighbor_grouping = _lib.SCHED_LOCALITY_NEIGHBOR_GROUPING
    nop = _lib.SCHED_NOP


REDUCE = {"sum": _lib.REDUCE_SUM, "mean": _lib.REDUCE_MEAN, "max": _lib.REDUCE_MAX}
MODE = {"rows": _lib.MODE_ROWS, "scheduled": _lib.MODE_SCHEDULED, "balanced": _lib.MODE_BALANCED}


def _dev_ptr(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise ValueError("%s must be a HIP device tensor (kernel.cpp:71 CHECK_CUDA)" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous (kernel.cpp:72 CHECK_CONTIGUOUS)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def _need_extras(what):
    """the backward entry points are out of scope and ship in libgnnagg_extras.so only"""
    if not _lib.has_extras():
        raise RuntimeError("%s: the loaded libgnnagg.so has no backward entry points (SURVEY.md 2.2: out of scope); build "
                           "`make -C gnn_computing_amd/csrc extras` and load it with GNNAGG_LIB=.../libgnnagg_extras.so" % what)


def _mode(scheduled):
    if isinstance(scheduled, str):
        return MODE[scheduled]
    return _lib.MODE_SCHEDULED if scheduled else _lib.MODE_ROWS


class Aggregator:
    """reference include/aggregator.h:25-151.  Holds the device CSR (borrowed tensors are kept
    alive by this object) and the scheduled work lists."""

    def __init__(self, ptr, idx, feat_in=32, feat_out=32):
        self.ptr, self.idx = ptr, idx
        self.num_v = int(ptr.numel()) - 1
        self.num_e = int(idx.numel())
        self.feat_in, self.feat_out = feat_in, feat_out
        self._h = ctypes.c_int64(0)

    # -- lifetime
    def close(self):
        if self._h.value:
            lib().gnnagg_destroy(self._h)
            self._h = ctypes.c_int64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _use_current_stream(self):
        """work of the following call is enqueued on torch's current stream; host-synchronous reads of the caller's arrays (plan
        construction) are ordered behind it too (gnnagg.h: copy_to_host)"""
        check(lib().gnnagg_set_stream(self._h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def set_option(self, name, value):
        """Per-handle knob (gnnagg_set_option): "partitions", "tile_width", "slice_kb", "fast_rows", ..."""
        check(lib().gnnagg_set_option(self._h, name.encode(), int(value)))

    # -- aggregator.h:67-99
    def schedule(self, s, param, total_num_v=None):
        arr = (ctypes.c_int * 2)(*(list(param) + [0])[:2])
        self._use_current_stream()   # (plan construction reads the CSR: ordered behind the stream that may still be writing it)
        check(lib().gnnagg_schedule(self._h, int(s), arr, self.num_v if total_num_v is None else int(total_num_v)))

    def plan_info(self):
        """{plan_s, rows_plan_s, plan_bytes, scratch_bytes} of the library-chosen blocked order (gnnagg_plan_info)."""
        a, b, pb, sb = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        check(lib().gnnagg_plan_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(pb), ctypes.byref(sb)))
        return {"plan_s": a.value, "rows_plan_s": b.value, "plan_bytes": pb.value, "scratch_bytes": sb.value}

    def schedule_balanced(self, chunk=0):
        self._use_current_stream()
        check(lib().gnnagg_schedule_balanced(self._h, int(chunk)))

    def balanced_params(self):
        """(chunk, seg_chunks) of the balanced mode's summation order (see gnnagg_balanced_params)."""
        ch, sg = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().gnnagg_balanced_params(self._h, ctypes.byref(ch), ctypes.byref(sg)))
        return ch.value, sg.value

    def rows_blocked_ranges(self):
        """0, or the number of source ranges when `scheduled = 0` runs its canonical chains on the 2-D blocked order
        (gnnagg_rows_blocked_ranges; GCN handles)."""
        n = ctypes.c_int(0)
        check(lib().gnnagg_rows_blocked_ranges(self._h, ctypes.byref(n)))
        return n.value

    def balanced_partitions(self):
        """0, or the number of source partitions when the balanced mode chose the partitioned order (gnnagg_balanced_partitions)."""
        n = ctypes.c_int(0)
        check(lib().gnnagg_balanced_partitions(self._h, ctypes.byref(n), None))
        return n.value

    def balanced_partition_columns(self):
        """Column count the source ranges are cut from (largest neighbor id + 1), 0 when not partitioned."""
        n, t = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().gnnagg_balanced_partitions(self._h, ctypes.byref(n), ctypes.byref(t)))
        return t.value

    def mode_params(self, mode="scheduled"):
        """(chunk, seg_chunks) of any mode's summation order (gnnagg_mode_params)."""
        ch, sg = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().gnnagg_mode_params(self._h, MODE[mode], ctypes.byref(ch), ctypes.byref(sg)))
        return ch.value, sg.value

    @property
    def num_target(self):
        """aggregator.h:126"""
        out = ctypes.c_int(0)
        check(lib().gnnagg_num_target(self._h, _lib.MODE_SCHEDULED, ctypes.byref(out)))
        return out.value

    def get_schedule(self, mode="scheduled", with_val=False):
        """Host copies (numpy) of ptr_s, idx_s, target[, val_s] as the kernels consume them."""
        m = MODE[mode]
        n = ctypes.c_int(0)
        check(lib().gnnagg_num_target(self._h, m, ctypes.byref(n)))
        ptr_s = np.empty(n.value + 1, np.int32)
        tgt = np.empty(n.value, np.int32)
        check(lib().gnnagg_get_schedule(self._h, m, ptr_s.ctypes.data, None, tgt.ctypes.data, None))
        ne = int(ptr_s[-1])
        idx_s = np.empty(ne, np.int32)
        val_s = np.empty(ne, np.float32) if with_val else None
        check(lib().gnnagg_get_schedule(self._h, m, None, idx_s.ctypes.data, None,
                                        val_s.ctypes.data if with_val else None))
        return (ptr_s, idx_s, tgt, val_s) if with_val else (ptr_s, idx_s, tgt)

    def check_csr(self, num_cols=0):
        """(rows with ptr[r] > ptr[r+1], neighbor ids outside [0, num_cols)) -- gnnagg_check_csr."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        self._use_current_stream()
        check(lib().gnnagg_check_csr(self._h, int(num_cols), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # -- aggregator.h:115-122
    def csr2edgelist(self):
        out = torch.empty(2 * self.num_e, dtype=torch.int32, device=self.ptr.device)
        self._use_current_stream()
        check(lib().gnnagg_csr2edgelist(self._h, ctypes.c_void_p(out.data_ptr())))
        return out


class Aggregator_GCN(Aggregator):
    """reference include/aggr_gcn.h:362-550"""

    def __init__(self, ptr, idx, val, feat_in=32, feat_out=32):
        super().__init__(ptr, idx, feat_in, feat_out)
        self.val = val
        check(lib().gnnagg_gcn_create(_dev_ptr(ptr, torch.int32, "ptr"), _dev_ptr(idx, torch.int32, "idx"),
                                      _dev_ptr(val, torch.float32, "val"), self.num_v, self.num_e,
                                      ctypes.byref(self._h)))

    def run(self, vin, vout, BLOCK_SIZE=512, scheduled=0, reduce="sum", accumulate=False, relu=False):
        """aggr_gcn.h:379-410.  BLOCK_SIZE is accepted for signature parity and ignored.
        accumulate=True (balanced mode, sum): vout += A.vin.  relu=True: vout = max(result, 0) in the same kernel
        (the F.relu that follows gcn_run in Figure7/our.py:176)."""
        return self.run_with_feat(vin, vout, BLOCK_SIZE, scheduled, int(vin.shape[1]), reduce, accumulate, relu)

    def run_with_feat(self, vin, vout, BLOCK_SIZE, scheduled, feat, reduce="sum", accumulate=False, relu=False):
        """aggr_gcn.h:411-444"""
        if vout.numel() < self.num_v * feat:
            raise ValueError("vout must hold num_v * feat floats")
        self.feat_in = feat
        self._use_current_stream()
        check(lib().gnnagg_gcn_run_ex(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vout, torch.float32, "vout"),
                                      int(feat), _mode(scheduled), REDUCE[reduce],
                                      (_lib.FLAG_ACCUMULATE if accumulate else 0) | (_lib.FLAG_RELU if relu else 0)))
        return 0.0

    def probe_gather(self, vin, scheduled="balanced"):
        """Measurement aid (gnnagg_gcn_probe_gather): the loads of run(vin, ., ., scheduled) without the FMA chains and
        without any store -- the gather ceiling of that launch."""
        self._use_current_stream()
        check(lib().gnnagg_gcn_probe_gather(self._h, _dev_ptr(vin, torch.float32, "vin"), int(vin.shape[1]), _mode(scheduled)))

    def run_clock(self, vin, vout, BLOCK_SIZE=64, scheduled=0):
        """aggr_gcn.h:462-489.  Returns an int64 tensor [blocks, 3] = (start tick, end tick, CU id) per workgroup of
        the one-item-per-lane-group kernel (ticks of gnnagg_wall_clock_hz())."""
        nb = ctypes.c_int(0)
        mode = _mode(scheduled)
        check(lib().gnnagg_gcn_run_clock(self._h, None, None, int(vin.shape[1]), mode, None, ctypes.byref(nb), None))
        timer = torch.zeros((max(nb.value, 1), 3), dtype=torch.int64, device=vin.device)
        self._use_current_stream()
        check(lib().gnnagg_gcn_run_clock(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vout, torch.float32, "vout"),
                                         int(vin.shape[1]), mode, ctypes.c_void_p(timer.data_ptr()), ctypes.byref(nb), None))
        return timer[:nb.value]

    def runEdgeWise(self, vin, vout, BLOCK_SIZE=512, scheduled=0):
        """aggr_gcn.h:446-460"""
        self._use_current_stream()
        check(lib().gnnagg_gcn_run_edgewise(self._h, _dev_ptr(vin, torch.float32, "vin"),
                                            _dev_ptr(vout, torch.float32, "vout"), int(vin.shape[1])))
        return 0.0

    def run_with_nn(self, vin, vout, weight, transformed, BLOCK_SIZE=128, scheduled=1):
        """aggr_gcn.h:491-499: vout = A.vin, transformed = vout @ weight (weight [feat_in, feat_out])."""
        self._use_current_stream()
        check(lib().gnnagg_gcn_run_with_nn(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vout, torch.float32, "vout"),
                                           _dev_ptr(weight, torch.float32, "weight"),
                                           _dev_ptr(transformed, torch.float32, "transformed"), int(vin.shape[1]),
                                           int(weight.shape[1]), _mode(scheduled)))

    def run_bwd(self, doutput, dinput):
        """d(input) = A^T . d(output) for the sum aggregation with this aggregator's edge values (extension: the reference
        is forward-only).  Deterministic gather over the transposed CSR."""
        _need_extras("Aggregator_GCN.run_bwd")
        self._use_current_stream()
        check(lib().gnnagg_gcn_run_bwd(self._h, _dev_ptr(doutput, torch.float32, "doutput"),
                                       _dev_ptr(dinput, torch.float32, "dinput"), int(doutput.shape[1])))

    def set_row_aux(self, row_aux):
        """gnnagg_set_row_aux: per-row degrees (int32 device tensor, or None) for means / maxima computed in two passes over
        disjoint edge sets -- the divisor of reduce="mean", the edges already folded into y for reduce="max" + accumulate."""
        self._row_aux = row_aux
        check(lib().gnnagg_set_row_aux(self._h, _dev_ptr(row_aux, torch.int32, "row_aux")))

    def updateval(self, val):
        """aggr_gcn.h:540-544"""
        self.val = val
        check(lib().gnnagg_update_val(self._h, _dev_ptr(val, torch.float32, "val")))


class Aggregator_GAT(Aggregator):
    """reference include/aggr_gat.h:299-441"""

    def __init__(self, ptr, idx, feat_in=32, feat_out=32):
        super().__init__(ptr, idx, feat_in, feat_out)
        check(lib().gnnagg_gat_create(_dev_ptr(ptr, torch.int32, "ptr"), _dev_ptr(idx, torch.int32, "idx"),
                                      self.num_v, self.num_e, ctypes.byref(self._h)))

    def run(self, vin, vatt, vout, BLOCK_SIZE=128, scheduled=0, heads=1, slope=0.2, newval=None):
        """aggr_gat.h:317-354 (slope 0.2 at :347); heads > 1 takes att [V,H,2]."""
        return self.run_with_feat(vin, vatt, vout, BLOCK_SIZE, scheduled, int(vin.shape[1]), heads, slope, newval)

    def run_with_feat(self, vin, vatt, vout, BLOCK_SIZE, scheduled, feat, heads=1, slope=0.2, newval=None):
        """aggr_gat.h:355-394"""
        if vatt.numel() < self.num_v * heads * 2:
            raise ValueError("att must hold at least V*heads*2 floats")
        self._use_current_stream()
        check(lib().gnnagg_gat_run(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vatt, torch.float32, "vatt"),
                                   _dev_ptr(vout, torch.float32, "vout"), int(feat), int(heads),
                                   ctypes.c_float(slope), _mode(scheduled), _dev_ptr(newval, torch.float32, "newval")))
        return 0.0

    def run_part(self, vin, vatt, vout, den_io, part, heads=1, slope=0.2):
        """gnnagg_gat_run_part: the fused aggregation in two passes over disjoint edge sets of the same rows.  part=1: vout
        receives the numerators, den_io [V, heads] the denominators; part=2: both are added to and the rows divided."""
        self._use_current_stream()
        check(lib().gnnagg_gat_run_part(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vatt, torch.float32, "vatt"),
                                        _dev_ptr(vout, torch.float32, "vout"), int(vin.shape[1]), int(heads), ctypes.c_float(slope),
                                        int(part), _dev_ptr(den_io, torch.float32, "den_io")))

    def probe_gather(self, vin, vatt, scheduled="balanced", heads=1):
        """Measurement aid (gnnagg_gat_probe_gather): the loads of run(vin, vatt, ., ., scheduled, heads) on the 2-D blocked
        order without exp, chains or stores -- the gather ceiling of that launch."""
        self._use_current_stream()
        check(lib().gnnagg_gat_probe_gather(self._h, _dev_ptr(vin, torch.float32, "vin"), _dev_ptr(vatt, torch.float32, "vatt"),
                                            int(vin.shape[1]), int(heads), _mode(scheduled)))

    def run_att(self, in_att, out_val, BLOCK_SIZE=128, heads=1, slope=0.2):
        """aggr_gat.h:395-401"""
        self._use_current_stream()
        check(lib().gnnagg_gat_run_att(self._h, _dev_ptr(in_att, torch.float32, "in_att"),
                                       _dev_ptr(out_val, torch.float32, "out_val"), int(heads), ctypes.c_float(slope)))

    def run_u_add_v(self, in_att, out_val, BLOCK_SIZE=128):
        """aggr_gat.h:402-409"""
        self._use_current_stream()
        check(lib().gnnagg_gat_run_u_add_v(self._h, _dev_ptr(in_att, torch.float32, "in_att"),
                                           _dev_ptr(out_val, torch.float32, "out_val")))

    def run_add_to_center(self, in_val, out_att, BLOCK_SIZE=128):
        """aggr_gat.h:410-417"""
        self._use_current_stream()
        check(lib().gnnagg_gat_run_add_to_center(self._h, _dev_ptr(in_val, torch.float32, "in_val"),
                                                 _dev_ptr(out_att, torch.float32, "out_att")))

    def run_div_each(self, in_att, in_out_val, BLOCK_SIZE=128):
        """aggr_gat.h:418-425"""
        self._use_current_stream()
        check(lib().gnnagg_gat_run_div_each(self._h, _dev_ptr(in_att, torch.float32, "in_att"),
                                            _dev_ptr(in_out_val, torch.float32, "in_out_val")))

    def run_bwd(self, output, doutput, newval, div, infeat, d_a_b, d_feat, relu_l=0.2, BLOCK_SIZE=128):
        """aggr_gat.h:426-434 (kernel :222-296).  Backward of the single-head fused aggregation from the forward pass's
        un-normalised edge weights `newval` [E] and denominators `div` [V]: d_feat [V,F] (through the aggregation) and
        d_a_b [V,2] (centre / source attention terms) are overwritten.  See gnnagg_gat_run_bwd for what the reference
        kernel leaves out."""
        self._use_current_stream()
        _need_extras("Aggregator_GAT.run_bwd")
        check(lib().gnnagg_gat_run_bwd(self._h, _dev_ptr(output, torch.float32, "output"),
                                       _dev_ptr(doutput, torch.float32, "doutput"), _dev_ptr(newval, torch.float32, "newval"),
                                       _dev_ptr(div, torch.float32, "div"), _dev_ptr(infeat, torch.float32, "infeat"),
                                       _dev_ptr(d_a_b, torch.float32, "d_a_b"), _dev_ptr(d_feat, torch.float32, "d_feat"),
                                       float(relu_l), int(infeat.shape[1])))


# ------------------------------------------------------------------------------------------
# Flat functions with the reference pybind names (Figure7/kernel.cpp:166-179).  Handles are
# Python objects here (the reference returns the raw pointer as int64 and leaks it).
# ------------------------------------------------------------------------------------------
def matmul_NN(A, B, C=None):
    """include/dense.h:4-23: row-major C = A @ B on the MFMA kernel of libgnnagg (device fp32 tensors)."""
    M, K = A.shape
    N = B.shape[1]
    if C is None:
        C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(lib().gnnagg_matmul_nn(_dev_ptr(A, torch.float32, "A"), _dev_ptr(B, torch.float32, "B"), _dev_ptr(C, torch.float32, "C"),
                                 int(M), int(N), int(K), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return C


def load_graph_host(dset, reorder="", datadir="../data/", shuffle=True):
    """gnnagg_load_graph -> dict of numpy arrays (ptr, idx, rows, reverse_rows)."""
    L = lib()
    nv, ne = ctypes.c_int(0), ctypes.c_int(0)
    P = _lib.P_INT
    pptr, pidx, prow, prrow = P(), P(), P(), P()
    check(L.gnnagg_load_graph(datadir.encode(), dset.encode(), reorder.encode(), int(bool(shuffle)), ctypes.byref(nv),
                              ctypes.byref(ne), ctypes.byref(pptr), ctypes.byref(pidx), ctypes.byref(prow),
                              ctypes.byref(prrow)))
    try:
        V, E = nv.value, ne.value
        out = dict(num_v=V, num_e=E,
                   ptr=np.ctypeslib.as_array(pptr, shape=(V + 1,)).copy(),
                   idx=np.ctypeslib.as_array(pidx, shape=(E,)).copy() if E else np.empty(0, np.int32),
                   rows=None, reverse_rows=None)
        if prow:
            out["rows"] = np.ctypeslib.as_array(prow, shape=(V,)).copy() if V else np.empty(0, np.int32)
            out["reverse_rows"] = np.ctypeslib.as_array(prrow, shape=(V,)).copy() if V else np.empty(0, np.int32)
    finally:
        for p in (pptr, pidx, prow, prrow):
            if p:
                L.gnnagg_free_host(ctypes.cast(p, ctypes.c_void_p))
    return out


def new_load(dset, reorder="", devid=0, datadir="../data/"):
    """kernel.cpp:37-67: returns [ptrs, idxs] as int32 device tensors."""
    g = load_graph_host(dset, reorder, datadir)
    dev = torch.device("cuda", devid)
    return [torch.from_numpy(g["ptr"]).to(dev), torch.from_numpy(g["idx"]).to(dev)]


def gcn_init(ptrs, idxs, val):
    """kernel.cpp:78-95.  Handles made through the reference-named surface start with the reference-facing defaults
    (gnnagg_set_option "reference_defaults"): gcn_run(..., scheduled=0) takes the balanced order -- within 1e-5 of the
    CSR-order chain instead of bit-equal to it; at.set_option("fast_rows", 0) restores the canonical chains."""
    at = Aggregator_GCN(ptrs, idxs, val)
    at.set_option("reference_defaults", 1)
    return at


def gcn_update_val(at, val):
    at.updateval(val)


def gcn_run(at, feat, outfeat, blocksize, scheduled, relu=False):
    """Figure7/kernel.cpp:97-106; relu=True fuses the F.relu that follows it in our.py:176 (extension)."""
    at.run_with_feat(feat, outfeat, blocksize, scheduled, int(feat.shape[1]), relu=relu)


def gcn_schedule(at, neighbor_num):
    at.schedule(Schedule.neighbor_grouping, [neighbor_num])


def gat_init(ptrs, idxs):
    at = Aggregator_GAT(ptrs, idxs)
    at.set_option("reference_defaults", 1)
    return at


def gat_run(at, feat, att, outfeat, blocksize, scheduled):
    at.run_with_feat(feat, att, outfeat, blocksize, scheduled, int(feat.shape[1]))


def gat_schedule(at, neighbor_num):
    at.schedule(Schedule.neighbor_grouping, [neighbor_num])


def gat_run_u_add_v(at, att, outval, blocksize):
    at.run_u_add_v(att, outval, blocksize)


def gat_run_add_to_center(at, inval, outatt, blocksize):
    at.run_add_to_center(inval, outatt, blocksize)


def gat_run_div_each(at, inatt, inoutval, blocksize):
    at.run_div_each(inatt, inoutval, blocksize)


# ------------------------------------------------------------------------------------------
# Host graph preparation through the C-ABI (numpy in / numpy out; no GPU needed)
# ------------------------------------------------------------------------------------------
def _np_i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def reorder_csr(ptr, idx, rows):
    """gnnagg_reorder_csr (reference src/data.cu:4-29).  Returns (newptr, newidx, reverse_rows)."""
    ptr, idx, rows = _np_i(ptr), _np_i(idx), _np_i(rows)
    V, E = len(ptr) - 1, len(idx)
    rev = np.empty(V, np.int32)
    rev[rows] = np.arange(V, dtype=np.int32)
    nptr, nidx = np.empty(V + 1, np.int32), np.empty(E, np.int32)
    check(lib().gnnagg_reorder_csr(ptr.ctypes.data, idx.ctypes.data, rows.ctypes.data, rev.ctypes.data, V, E,
                                   nptr.ctypes.data, nidx.ctypes.data))
    return nptr, nidx, rev


def neighbor_grouping_schedule(ptr, ng):
    """gnnagg_neighbor_grouping_schedule (reference graph_schedule.h:91-126) -> (ptr_s, target)."""
    ptr = _np_i(ptr)
    V = len(ptr) - 1
    n = ctypes.c_int(0)
    check(lib().gnnagg_neighbor_grouping_schedule(ptr.ctypes.data, int(ng), V, None, None, ctypes.byref(n)))
    ptr_s, tgt = np.empty(n.value + 1, np.int32), np.empty(n.value, np.int32)
    check(lib().gnnagg_neighbor_grouping_schedule(ptr.ctypes.data, int(ng), V, ptr_s.ctypes.data, tgt.ctypes.data,
                                                  ctypes.byref(n)))
    return ptr_s, tgt


def locality_schedule(ptr, idx, par_num, total_v, ng=0, val=None):
    """gnnagg_locality_schedule (reference graph_schedule.h:17-63 / :156-211)."""
    ptr, idx = _np_i(ptr), _np_i(idx)
    V, E = len(ptr) - 1, len(idx)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    ptr_s, idx_s, tgt = np.empty(E + 2, np.int32), np.empty(max(E, 1), np.int32), np.empty(max(E, 1), np.int32)
    val_s = np.empty(max(E, 1), np.float32) if val is not None else None
    n = ctypes.c_int(0)
    check(lib().gnnagg_locality_schedule(ptr.ctypes.data, idx.ctypes.data, None if val is None else val.ctypes.data,
                                         int(par_num), int(ng), V, int(total_v), ptr_s.ctypes.data, idx_s.ctypes.data,
                                         None if val is None else val_s.ctypes.data, tgt.ctypes.data, ctypes.byref(n)))
    G = n.value
    ne = int(ptr_s[G])
    return ptr_s[:G + 1].copy(), idx_s[:ne].copy(), tgt[:G].copy(), (None if val is None else val_s[:ne].copy())


def cluster_reorder(ptr, idx, threshold=0.2, num_perm=64, cluster_cap=64, seed=123, order="first_member", cache_rows=4096):
    """gnnagg_cluster_reorder[_ex] (reference script/cluster2.py).  Returns (rows, num_clusters); rows[i] = old node id
    placed at new position i, ready for graph.write_reorder_file / reorder_csr.  order="first_member" writes the clusters
    as the reference script does; order="cache_greedy" orders them with an LRU model of `cache_rows` feature rows."""
    ptr, idx = _np_i(ptr), _np_i(idx)
    V = len(ptr) - 1
    rows = np.empty(V, np.int32)
    nc = ctypes.c_int(0)
    mode = {"first_member": 0, "cache_greedy": 1}[order]
    check(lib().gnnagg_cluster_reorder_ex(ptr.ctypes.data, idx.ctypes.data, V, ctypes.c_float(threshold), int(num_perm),
                                          int(cluster_cap), ctypes.c_ulonglong(seed), mode, int(cache_rows), rows.ctypes.data,
                                          ctypes.byref(nc)))
    return rows, nc.value


def partition_rows(ptr, nparts):
    ptr = _np_i(ptr)
    b = np.empty(nparts + 1, np.int32)
    check(lib().gnnagg_partition_rows(ptr.ctypes.data, len(ptr) - 1, int(nparts), b.ctypes.data))
    return b


def halo_plan(ptr, idx, bounds, rank, row_slice=False, num_cols=None):
    """gnnagg_halo_plan / gnnagg_halo_plan_slice -> dict(local_ptr, local_idx, halo_ids, halo_counts).  row_slice=True:
    (ptr, idx) hold the rank's own rows only (ptr[0 .. n_local] with any base, global column ids in idx)."""
    ptr, idx, bounds = _np_i(ptr), _np_i(idx), _np_i(bounds)
    nparts = len(bounds) - 1
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    if row_slice:
        if len(ptr) != r1 - r0 + 1:
            raise ValueError("row slice must have bounds[rank+1] - bounds[rank] + 1 ptr entries")
        nnz = int(ptr[-1] - ptr[0])
    else:
        nnz = int(ptr[r1] - ptr[r0])
    lptr, lidx = np.empty(r1 - r0 + 1, np.int32), np.empty(max(nnz, 1), np.int32)
    counts = np.empty(nparts, np.int32)
    ids_p, nh = _lib.P_INT(), ctypes.c_int(0)
    if row_slice:
        check(lib().gnnagg_halo_plan_slice(ptr.ctypes.data, idx.ctypes.data, int(num_cols), bounds.ctypes.data, nparts, int(rank),
                                           lptr.ctypes.data, lidx.ctypes.data, ctypes.byref(ids_p), counts.ctypes.data,
                                           ctypes.byref(nh)))
    else:
        check(lib().gnnagg_halo_plan(ptr.ctypes.data, idx.ctypes.data, len(ptr) - 1, bounds.ctypes.data, nparts, int(rank),
                                     lptr.ctypes.data, lidx.ctypes.data, ctypes.byref(ids_p), counts.ctypes.data,
                                     ctypes.byref(nh)))
    ids = np.ctypeslib.as_array(ids_p, shape=(nh.value,)).copy() if nh.value else np.empty(0, np.int32)
    lib().gnnagg_free_host(ctypes.cast(ids_p, ctypes.c_void_p))
    return dict(local_ptr=lptr, local_idx=lidx[:nnz].copy(), halo_ids=ids, halo_counts=counts, n_local=r1 - r0)
