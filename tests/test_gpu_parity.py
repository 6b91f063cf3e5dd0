"""GPU parity tests: the HIP path (through the C-ABI of libgnnagg.so) against the CPU oracle.

Bar (north_star): bit-exact for index/degree work; fp32 aggregation within 1e-5 relative, stated as
the condition-aware bound |y - y_ref| <= 1e-5 * sum_e |val_e * x_e| -- and in fact
bit-exact wherever the kernel keeps the oracle's summation order, which these tests assert.
"""
import ctypes

import numpy as np
import pytest
import torch

import gnn_computing_amd as gnc
from gnn_computing_amd import _lib
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RTOL = 1e-5

# the backward entry points are out of scope and ship in libgnnagg_extras.so only: second tier,
# GNNAGG_LIB=gnn_computing_amd/libgnnagg_extras.so GNNAGG_TEST_TIER=2
from gnn_computing_amd import _lib as _gl  # noqa: E402
needs_extras = pytest.mark.skipif(not _gl.has_extras(), reason="second tier: needs libgnnagg_extras.so (GNNAGG_LIB)")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_graph(V, E, seed, sorted_rows=False):
    ptr, idx = gnc.graph.uniform_random_csr(V, E, seed)
    if sorted_rows:
        for r in range(V):
            idx[ptr[r]:ptr[r + 1]].sort()
    return ptr, idx


def rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def assert_within(y, ref, scale, what):
    """|y - ref| <= RTOL * scale elementwise, scale = sum_e |val_e x_e| (plus a floor of one ulp of it)."""
    err = np.abs(y.astype(np.float64) - ref.astype(np.float64))
    bound = RTOL * scale.astype(np.float64) + 1e-30
    bad = err > bound
    assert not bad.any(), "%s: %d elements outside 1e-5*sum|v x| (worst ratio %.3g)" % (
        what, int(bad.sum()), float((err / bound).max()))


CASES = [  # (V, E, F)
    (1, 0, 4), (1, 3, 128), (7, 0, 32), (50, 400, 1), (50, 400, 3), (64, 900, 32), (200, 3000, 64),
    (300, 5000, 100), (257, 4000, 128), (129, 3000, 256), (90, 2500, 602), (40, 800, 1024),
]


@pytest.mark.parametrize("V,E,F", CASES)
@pytest.mark.parametrize("with_val", [True, False])
def test_gcn_rows_bit_exact(V, E, F, with_val):
    ptr, idx = make_graph(V, E, seed=V * 7 + F)
    x, val = rand((V, F), 1), (rand(E, 2) if with_val else None)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if val is None else dev(val), F, F)
    y = torch.full((V, F), 7.0, device=DEV)  # poison: every element must be overwritten
    agg.run(dev(x), y, 512, 0)
    ref = orc.gcn_seq(ptr, idx, val, x)
    assert np.array_equal(y.cpu().numpy(), ref)


@pytest.mark.parametrize("V,E,F", [(64, 900, 32), (257, 6000, 128), (300, 5000, 100), (90, 2500, 602)])
@pytest.mark.parametrize("ng", [1, 2, 16, 32])
def test_gcn_neighbor_grouping_bit_exact(V, E, F, ng):
    ptr, idx = make_graph(V, E, seed=V + ng)
    x, val = rand((V, F), 3), rand(E, 4)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [ng])
    ps_ref, tg_ref = orc.neighbor_grouping(ptr, ng)
    ps, ix, tg = agg.get_schedule("scheduled")
    assert agg.num_target == len(tg_ref)
    assert np.array_equal(ps, ps_ref) and np.array_equal(tg, tg_ref) and np.array_equal(ix, idx)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 512, 1)
    chunk, seg = agg.mode_params("scheduled")  # (NG, 16) on the plan kernel, (NG, 0) on the item kernel (tiny NG)
    assert chunk == ng
    ref = orc.gcn_grouped(ps_ref, tg_ref, idx, val, x, V, seg=seg)
    assert np.array_equal(y.cpu().numpy(), ref)
    # and against the canonical CSR-order chain within the fp32 bound
    assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x), "NG vs CSR order")


@pytest.mark.parametrize("kind,param", [("locality", [3]), ("locality_neighbor_grouping", [3, 4]),
                                        ("locality", [1]), ("locality_neighbor_grouping", [7, 1])])
def test_gcn_locality_schedules(kind, param):
    V, E, F = 150, 2500, 64
    ptr, idx = make_graph(V, E, seed=11)
    x, val = rand((V, F), 5), rand(E, 6)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule[kind], param)
    ng = param[1] if len(param) > 1 else 0
    ps_ref, ix_ref, tg_ref, vs_ref = orc.locality_schedule(ptr, idx, param[0], V, ng, val)
    ps, ix, tg, vs = agg.get_schedule("scheduled", with_val=True)
    assert np.array_equal(ps, ps_ref) and np.array_equal(ix, ix_ref) and np.array_equal(tg, tg_ref)
    assert np.array_equal(vs, vs_ref)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 512, 1)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps_ref, tg_ref, ix_ref, vs_ref, x, V))


@pytest.mark.parametrize("F", [32, 128, 100])
def test_gcn_balanced_mode(F):
    V, E = 400, 30000  # average degree 75 -> several chunks per row
    ptr, idx = make_graph(V, E, seed=21)
    x, val = rand((V, F), 7), rand(E, 8)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 512, "balanced")
    ps, ix, tg = agg.get_schedule("balanced")
    chunk, seg = agg.balanced_params()
    assert np.array_equal(ps, orc.neighbor_grouping(ptr, chunk)[0])
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=seg))
    assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x), "balanced")
    agg.schedule_balanced(8)  # rows of 8..128 edges: one segment; longer rows: several segments + combine
    agg.run(dev(x), y, 512, "balanced")
    ps, ix, tg = agg.get_schedule("balanced")
    chunk, seg = agg.balanced_params()
    assert chunk == 8 and np.array_equal(ps, orc.neighbor_grouping(ptr, 8)[0])
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=seg))
    for red, ref in (("max", orc.gcn_max(ptr, idx, val, x)),):
        agg.run(dev(x), y, 512, "balanced", reduce=red)
        assert np.array_equal(y.cpu().numpy(), ref)
    agg.run(dev(x), y, 512, "balanced", reduce="mean")
    d = np.maximum(orc.degrees(ptr), 1)[:, None].astype(np.float32)
    assert_within(y.cpu().numpy(), orc.gcn_mean(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x) / d, "balanced mean")


@pytest.mark.parametrize("mode", ["rows", "scheduled"])
@pytest.mark.parametrize("F", [32, 128, 602])
def test_gcn_mean_max(mode, F):
    V, E = 120, 3000
    ptr, idx = make_graph(V, E, seed=31)
    x, val = rand((V, F), 9), rand(E, 10)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [4])
    y = torch.full((V, F), 7.0, device=DEV)
    sched = mode == "scheduled"
    agg.run(dev(x), y, 512, sched, reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))  # max is order-independent
    agg.run(dev(x), y, 512, sched, reduce="mean")
    ref = orc.gcn_mean(ptr, idx, val, x)
    if not sched:
        assert np.array_equal(y.cpu().numpy(), ref)
    else:
        deg = np.maximum(orc.degrees(ptr), 1)[:, None].astype(np.float32)
        assert_within(y.cpu().numpy(), ref, orc.gcn_abs_scale(ptr, idx, val, x) / deg, "mean scheduled")
    # implicit weights (GraphSAGE mean: val = None)
    agg2 = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    agg2.run(dev(x), y, 512, 0, reduce="mean")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_mean(ptr, idx, None, x))


def test_gcn_update_val_aliases():
    V, E, F = 100, 1500, 64
    ptr, idx = make_graph(V, E, seed=41)
    x, v1, v2 = rand((V, F), 1), rand(E, 2), rand(E, 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(v1), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [16])
    agg.updateval(dev(v2))  # aggr_gcn.h:540-544: both the CSR and the scheduled val follow
    y = torch.empty((V, F), device=DEV)
    agg.run(dev(x), y, 128, 0)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, v2, x))
    agg.run(dev(x), y, 128, 1)
    ps, tg = orc.neighbor_grouping(ptr, 16)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, v2, x, V, seg=agg.mode_params("scheduled")[1]))


def test_csr2edgelist_and_edgewise():
    V, E, F = 130, 2000, 48
    ptr, idx = make_graph(V, E, seed=51)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    assert np.array_equal(agg.csr2edgelist().cpu().numpy(), orc.csr2edgelist(ptr, idx))
    y = torch.full((V, F), 7.0, device=DEV)
    agg.runEdgeWise(dev(x), y)
    # one product per edge added atomically in arbitrary order: fp32 bound only
    assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x), "edgewise")


def test_spmm_naive_and_validators():
    V, E, F = 140, 2200, 40
    ptr, idx = make_graph(V, E, seed=61)
    x, val = rand((V, F), 1), rand(E, 2)
    y0 = rand((V, F), 3)
    y = dev(y0)
    L = gnc.lib()
    dptr, didx, dval, dx = dev(ptr), dev(idx), dev(val), dev(x)
    _lib.check(L.gnnagg_spmm_naive(dptr.data_ptr(), didx.data_ptr(), dval.data_ptr(), dx.data_ptr(), y.data_ptr(), V, F,
                                   None))
    torch.cuda.synchronize()
    ref = orc.spmm_naive(ptr, idx, val, x, y0)  # empty rows keep y0 (spmm.h:236-237)
    assert np.array_equal(y.cpu().numpy(), ref)
    # validators: perturb some elements
    ans = ref.copy()
    ans[::7, ::5] *= 1.05
    n = ctypes.c_int(-1)
    d_ref, d_ans = dev(ref), dev(ans)  # keep the tensors alive across the call
    _lib.check(L.gnnagg_validate(d_ref.data_ptr(), d_ans.data_ptr(), ref.size, ctypes.byref(n), None))
    assert n.value == orc.validate2(ref, ans)
    rows = np.random.default_rng(5).permutation(V).astype(np.int32)
    ans_perm = np.empty_like(ref)
    ans_perm[rows] = ref  # ans[map[r]] = ref[r]
    ans_perm[3] += 1.0
    d_perm, d_rows = dev(ans_perm), dev(rows)
    _lib.check(L.gnnagg_validate_reordered(d_ref.data_ptr(), d_perm.data_ptr(), d_rows.data_ptr(), V, F,
                                           ctypes.byref(n), None))
    assert n.value == orc.validate_reordered(ref, ans_perm, rows) == F


# ------------------------------------------------------------------------------------- GAT
def gat_scale(ptr, idx, att, x, heads, slope=0.2):
    """sum_e w_e |x_e| / sum_e w_e : the error scale of the normalised output."""
    w = orc.gat_att(ptr, idx, att, heads, slope)  # normalised weights [E,H]
    V, F = len(ptr) - 1, x.shape[1]
    s = np.zeros((V, F))
    if len(idx):
        rows = np.repeat(np.arange(V), np.diff(ptr))
        np.add.at(s, rows, np.repeat(w, F // heads, axis=1).astype(np.float64) * np.abs(x[idx]))
    return s.astype(np.float32)


@pytest.mark.parametrize("V,E,F,H", [(60, 700, 32, 1), (150, 2500, 128, 1), (80, 1500, 256, 8), (70, 900, 96, 4),
                                     (50, 600, 30, 3), (9, 0, 32, 1)])
def test_gat_fused_rows(V, E, F, H):
    ptr, idx = make_graph(V, E, seed=71 + F)
    x, att = rand((V, F), 1), rand((V, H, 2), 2)
    agg = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), dev(att), y, 128, 0, heads=H)
    ref = orc.gat_fused(ptr, idx, att, x, H)
    # expf on device vs libm: a few ulp on each w_e -> relative 1e-5 of the weighted magnitude
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "gat fused")
    assert np.all(y.cpu().numpy()[orc.degrees(ptr) == 0] == 0)  # empty rows -> 0, not NaN


@pytest.mark.parametrize("F,H,ng", [(32, 1, 32), (128, 1, 16), (256, 8, 32), (64, 2, 3)])
def test_gat_scheduled(F, H, ng):
    V, E = 120, 4000
    ptr, idx = make_graph(V, E, seed=81)
    x, att = rand((V, F), 1), rand((V, H, 2), 2)
    agg = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [ng])
    y = torch.full((V, F), 7.0, device=DEV)
    newval = torch.full((E, H), 7.0, device=DEV)
    agg.run(dev(x), dev(att), y, 128, 1, heads=H, newval=newval)
    ps, tg = orc.neighbor_grouping(ptr, ng)
    ref, ref_newval, _ = orc.gat_grouped(ps, tg, idx, att, x, V, H, seg=agg.mode_params("scheduled")[1])
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "gat scheduled")
    np.testing.assert_allclose(newval.cpu().numpy(), ref_newval, rtol=1e-6)
    # the fused result agrees with the unscheduled one (same math, different association)
    y2 = torch.empty_like(y)
    agg.run(dev(x), dev(att), y2, 128, 0, heads=H)
    assert_within(y.cpu().numpy(), y2.cpu().numpy(), gat_scale(ptr, idx, att, x, H) + np.abs(ref), "gat sched vs rows")


@pytest.mark.parametrize("H", [1, 8])
def test_gat_adapter_and_three_step(H):
    V, E, F = 140, 2600, 32 * H
    ptr, idx = make_graph(V, E, seed=91)
    x, att = rand((V, F), 1), rand((V, H, 2), 2)
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    out_val = torch.full((E, H), 7.0, device=DEV)
    gat.run_att(dev(att), out_val, 128, heads=H)
    np.testing.assert_allclose(out_val.cpu().numpy(), orc.gat_att(ptr, idx, att, H), rtol=RTOL)
    if H == 1:
        # adapter path of Figure10/main_a.cu:98-100: run_att -> updateval -> gcn.run == fused gat.run
        gcn = gnc.Aggregator_GCN(dev(ptr), dev(idx), out_val.view(-1), F, F)
        y = torch.empty((V, F), device=DEV)
        gcn.run(dev(x), y, 128, 0)
        ref = orc.gat_fused(ptr, idx, att, x, 1)
        assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "adapter")
        # 3-step baseline (aggr_gat.h:33-92): u_add_v, caller-side exp(leaky_relu), add_to_center, div_each
        v = torch.empty(E, device=DEV)
        gat.run_u_add_v(dev(att), v)
        assert np.array_equal(v.cpu().numpy(), orc.gat_u_add_v(ptr, idx, att))
        v = torch.exp(torch.nn.functional.leaky_relu(v, 0.2))
        center = torch.full((V,), 7.0, device=DEV)
        gat.run_add_to_center(v, center)
        np.testing.assert_allclose(center.cpu().numpy(), orc.gat_add_to_center(ptr, v.cpu().numpy()), rtol=RTOL)
        vh = v.cpu().numpy()
        gat.run_div_each(center, v)
        assert np.array_equal(v.cpu().numpy(), orc.gat_div_each(ptr, center.cpu().numpy(), vh))


# ------------------------------------------------------------------------- boundary behaviour
def test_error_behaviour():
    V, E, F = 20, 100, 32
    ptr, idx = make_graph(V, E, seed=5)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    x, y = dev(rand((V, F), 1)), torch.empty((V, F), device=DEV)
    with pytest.raises(gnc.GnnAggError) as ei:
        agg.run(x, y, 512, 1)  # scheduled run without schedule(): reference asserts (aggr_gcn.h:392)
    assert ei.value.code == _lib.ERR_STATE
    with pytest.raises(gnc.GnnAggError):
        agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg.schedule(gnc.Schedule.neighbor_grouping, [0])
    with pytest.raises(ValueError):
        agg.run(x.cpu(), y, 512, 0)
    h = agg._h.value
    agg.close()
    assert gnc.lib().gnnagg_destroy(ctypes.c_int64(h)) == _lib.ERR_ARG  # double destroy is caught
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    with pytest.raises(gnc.GnnAggError):
        gat.run(x, dev(rand((V, 3, 2), 2)), y, 128, 0, heads=3)  # 32 % 3 != 0


@pytest.mark.null_stream
def test_flat_reference_api():
    """The Section-A entry points (reference Figure7/kernel.cpp:15-35), called the way the reference's
    kernel.cpp wrappers call them: raw device pointers + sizes."""
    V, E, F = 90, 1400, 128
    ptr, idx = make_graph(V, E, seed=15)
    x, val, att = rand((V, F), 1), np.ones(E, np.float32), rand((V, 2), 3)
    dptr, didx, dval, dx, datt = dev(ptr), dev(idx), dev(val), dev(x), dev(att)
    y = torch.full((V, F), 7.0, device=DEV)
    L = gnc.lib()
    at = L.GCN_init_impl(dptr.data_ptr(), didx.data_ptr(), dval.data_ptr(), V, E)
    assert at != 0
    arr = (ctypes.c_int * 1)(32)
    L.GCN_schedule_impl(at, arr)
    # defaults of the reference-facing surface: scheduled = 0 and scheduled = 1 both run the balanced order (the reference's
    # scheduled kernel adds with atomics, any association is one of its results; scheduled = 0 differs from aggr_gcn's chain by
    # association only) -- bit-equal to the restatement of THAT order, within the 1e-5 bound of the sequential chain
    n = ctypes.c_int(0)
    _lib.check(L.gnnagg_num_target(at, _lib.MODE_BALANCED, ctypes.byref(n)))
    bps, btg = np.empty(n.value + 1, np.int32), np.empty(n.value, np.int32)
    _lib.check(L.gnnagg_get_schedule(at, _lib.MODE_BALANCED, bps.ctypes.data, None, btg.ctypes.data, None))
    bch, bsg = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.gnnagg_balanced_params(at, ctypes.byref(bch), ctypes.byref(bsg)))
    y_bal = orc.gcn_grouped(bps, btg, idx, val, x, V, seg=bsg.value)
    y_seq, scale = orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x)
    for sched in (1, 0):
        y.fill_(7.0)
        L.GCN_run_impl(at, dx.data_ptr(), y.data_ptr(), 128, sched, F)
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), y_bal), "flat API default, scheduled = %d" % sched
        assert_within(y.cpu().numpy(), y_seq, scale, "flat API default vs the sequential chain")
    # num_target / get_schedule / mode_params of the scheduled mode keep describing the user's groups
    ps, tg = orc.neighbor_grouping(ptr, 32)
    _lib.check(L.gnnagg_num_target(at, _lib.MODE_SCHEDULED, ctypes.byref(n)))
    assert n.value == len(tg)
    # the canonical orders, one option away
    _lib.check(L.gnnagg_set_option(at, b"fast_scheduled", 0))
    _lib.check(L.gnnagg_set_option(at, b"fast_rows", 0))
    L.GCN_run_impl(at, dx.data_ptr(), y.data_ptr(), 128, 1, F)
    torch.cuda.synchronize()
    ch, sg = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.gnnagg_mode_params(at, _lib.MODE_SCHEDULED, ctypes.byref(ch), ctypes.byref(sg)))
    assert ch.value == 32
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=sg.value))
    L.GCN_run_impl(at, dx.data_ptr(), y.data_ptr(), 128, 0, F)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x))
    v2 = dev(rand(E, 9))
    L.GCN_update_val_impl(at, v2.data_ptr())
    L.GCN_run_impl(at, dx.data_ptr(), y.data_ptr(), 128, 0, F)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, v2.cpu().numpy(), x))
    g = L.GAT_init_impl(dptr.data_ptr(), didx.data_ptr(), V, E)
    L.GAT_schedule_impl(g, arr)
    for sched in (0, 1):
        L.GAT_run_impl(g, dx.data_ptr(), datt.data_ptr(), y.data_ptr(), 128, sched, F)
        torch.cuda.synchronize()
        ref = orc.gat_fused(ptr, idx, att, x, 1)
        assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "flat gat")
    ev = torch.empty(E, device=DEV)
    L.GAT_run_u_add_v_impl(g, datt.data_ptr(), ev.data_ptr(), 128)
    torch.cuda.synchronize()
    assert np.array_equal(ev.cpu().numpy(), orc.gat_u_add_v(ptr, idx, att))
    assert L.gnnagg_destroy(at) == 0 and L.gnnagg_destroy(g) == 0


def test_stream_is_honoured():
    V, E, F = 300, 6000, 128
    ptr, idx = make_graph(V, E, seed=25)
    x = rand((V, F), 1)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    dx, y = dev(x), torch.empty((V, F), device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        agg.run(dx, y, 512, 0)
    s.synchronize()
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, None, x))


def test_pack_rows():
    n, V, F = 77, 200, 100
    x = rand((V, F), 1)
    ids = np.random.default_rng(3).integers(0, V, n).astype(np.int32)
    out = torch.empty((n, F), device=DEV)
    d_x, d_ids = dev(x), dev(ids)
    _lib.check(gnc.lib().gnnagg_pack_rows(d_x.data_ptr(), d_ids.data_ptr(), n, F, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), x[ids])


# --------------------------------------------------------------------------- full-size checks
def test_arxiv_full_size_parity_and_properties():
    """BASELINE configs[1]: arxiv-shaped CSR (169 343 x 1 166 243), GCN sum, F=128.  The oracle finishes
    this size in well under a second, so parity is checked directly, plus size-independent properties."""
    ptr_t, idx_t = gnc.graph.dataset("arxiv")
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    V, E, F = len(ptr) - 1, len(idx), 128
    assert (V, E) == gnc.graph.SHAPES["arxiv"]
    x, val = rand((V, F), 123), np.ones(E, np.float32)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    dx, y = dev(x), torch.empty((V, F), device=DEV)
    agg.run(dx, y, 512, 0)
    ref = orc.gcn_seq(ptr, idx, val, x)
    assert np.array_equal(y.cpu().numpy(), ref)
    agg.run(dx, y, 512, "balanced")
    ps, ix, tg = agg.get_schedule("balanced")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.balanced_params()[1]))
    assert_within(y.cpu().numpy(), ref, orc.gcn_abs_scale(ptr, idx, val, x), "arxiv balanced")
    # degree property: X = ones -> every column equals the in-degree (exact in fp32 below 2^24)
    ones = torch.ones((V, F), device=DEV)
    agg.run(ones, y, 512, "balanced")
    assert np.array_equal(y.cpu().numpy()[:, 0], orc.degrees(ptr).astype(np.float32))
    assert float(y.sum().item()) == float(E) * F
    # reorder invariance: permuting the graph permutes the output rows (validReordered's contract)
    rows = np.random.default_rng(7).permutation(V).astype(np.int32)
    nptr, nidx, rev = gnc.reorder_csr(ptr, idx, rows)
    agg2 = gnc.Aggregator_GCN(dev(nptr), dev(nidx), dev(val), F, F)
    y2 = torch.empty((V, F), device=DEV)
    agg2.run(dev(x[rows]), y2, 512, 0)
    n = ctypes.c_int(-1)
    agg.run(dx, y, 512, 0)
    # y (old order) row r must equal y2 row reverse_rows[r]
    d_rev = dev(rev)
    _lib.check(gnc.lib().gnnagg_validate_reordered(y.data_ptr(), y2.data_ptr(), d_rev.data_ptr(), V, F,
                                                   ctypes.byref(n), None))
    assert n.value == 0
    assert np.array_equal(y2.cpu().numpy()[rev], y.cpu().numpy())


@pytest.mark.parametrize("F,ng", [(128, 2), (100, 7), (602, 16), (32, 1)])
def test_hub_rows_block_cooperative_combine(F, ng):
    """Hub rows (thousands of partial rows) go through the workgroup-per-row combine path."""
    V = 300
    rng = np.random.default_rng(5)
    deg = rng.integers(0, 6, V)
    deg[7], deg[150], deg[299] = 5000, 1337, 18  # two hubs, one row just above the batch size
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [ng])
    ps, tg = orc.neighbor_grouping(ptr, ng)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 512, 1)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.mode_params("scheduled")[1]))
    agg.run(dev(x), y, 512, 1, reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    # rows mode keeps the canonical CSR-order chain even for the hubs (workgroup-per-row path), all reductions
    agg.run(dev(x), y, 512, 0)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x))
    agg.run(dev(x), y, 512, 0, reduce="mean")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_mean(ptr, idx, val, x))
    agg.run(dev(x), y, 512, 0, reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    # the balanced plan with the same chunk: hubs span several 16-chunk segments (+ combine)
    agg.schedule_balanced(ng)
    agg.run(dev(x), y, 512, "balanced")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.balanced_params()[1]))
    agg.run(dev(x), y, 512, "balanced", reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    agg.run(dev(x), y, 512, 1, reduce="mean")
    d = np.maximum(deg, 1)[:, None].astype(np.float32)
    assert_within(y.cpu().numpy(), orc.gcn_mean(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x) / d, "hub mean")
    if F % 2 == 0 and F <= 128:
        H = 2
        att = rand((V, H, 2), 3) * 0.3
        gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        gat.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        gat.schedule(gnc.Schedule.neighbor_grouping, [ng])
        gat.run(dev(x), dev(att), y, 128, 1, heads=H)
        ref, _, _ = orc.gat_grouped(ps, tg, idx, att, x, V, H, seg=gat.mode_params("scheduled")[1])
        assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "hub gat scheduled")


def test_cpp_drivers_run(tmp_path):
    """The C++ class shim (include/compat/) + drivers with the reference's flags and call sequence (Figure9/main.cu:59-74,
    Figure10/main_a.cu:82-110, main_b.cu:86-103) run end to end on a small dataset directory -- and what they COMPUTED is checked, not only
    that they exit 0: `--dump DIR` makes the shim write the operands of the last call of every entry point
    (GNNAGG_COMPAT_DUMP, include/compat/util.h) and tests/driver_dumps.py compares them with the oracle at the suite's bounds."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import driver_dumps as dd
    d = str(tmp_path) + "/"
    ptr_t, idx_t = gnc.graph.powerlaw_csr(5000, 60000, seed=3)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    gnc.graph.write_graph_files(d, "tiny", ptr, idx, text=True)
    rows = np.random.default_rng(1).permutation(5000).astype(np.int32)
    gnc.graph.write_reorder_file(d, "tiny", rows)
    rptr, ridx, _, _ = orc.reorder_csr(ptr, idx, rows)
    F = 64
    cases = (("fig9.out", ["--reorder", "_thres_0.2"], {}), ("fig9.out", [], {"GNNAGG_FAST_ROWS": "0"}), ("fig10a.out", ["--nei", "32"], {}),
             ("fig10b.out", ["--outfea", "32"], {}), ("fig8.out", ["--nei", "16"], {}))
    for k, (exe, extra, env) in enumerate(cases):
        path = os.path.join(root, "drivers", exe)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(root, "drivers")])
        dump = os.path.join(d, "dump%d" % k)
        os.makedirs(dump)
        r = subprocess.run([path, "--dataset", "tiny", "--datadir", d, "--feature-len", str(F), "--dump", dump] + extra,
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
        assert len(lines) >= 2 and all(l.get("seconds", l.get("actual_seconds", 0)) > 0 for l in lines)
        p, i = (rptr, ridx) if "--reorder" in extra else (ptr, idx)
        if exe == "fig9.out":
            # run(x, y, B, 0): the balanced order by default (inside the bound), aggr_gcn's own chain -- bit-equal -- with GNNAGG_FAST_ROWS=0
            dd.gcn(dump, "gcn_run_s0", p, i, F, exact=env.get("GNNAGG_FAST_ROWS") == "0")
            dd.gcn(dump, "gcn_run_s1", p, i, F)
            dd.gcn(dump, "gcn_run_balanced", p, i, F)
        elif exe == "fig10a.out":
            dd.edge_softmax_stages(dump, p, i, with_exp=True)
            dd.gcn(dump, "gcn_run_s1", p, i, F)            # the adapter: weighted SpMM with run_att's values (aggr_gcn.h:540-544)
            w = dd.read(dump, "gcn_run_s1", "val")
            np.testing.assert_allclose(w, dd.read(dump, "gat_run_att", "val"), rtol=0, atol=0)
            dd.gat(dump, "gat_run_s1", p, i, F)
            dd.gat(dump, "gat_run_heads", p, i, F)
            # adapter == fused (Figure 10a's claim), both inside the bound of the same edge softmax
            ya, yf = dd.read(dump, "gcn_run_s1", "y"), dd.read(dump, "gat_run_s1", "y")
            assert np.allclose(ya, yf, rtol=1e-4, atol=1e-5)
        elif exe == "fig10b.out":
            dd.gcn(dump, "gcn_run_s1", p, i, F)
            dd.matmul(dump, M=5000, K=F, N=32)
            dd.run_with_nn(dump, p, i, F, 32)


@pytest.mark.parametrize("H", [1, 3, 4])
def test_edge_softmax_kernels_with_hub_rows(H):
    """run_att / u_add_v / add_to_center / div_each on chunked work items: hub rows span many items."""
    V = 400
    rng = np.random.default_rng(9)
    deg = rng.integers(0, 5, V)
    deg[3], deg[200] = 3000, 700
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    att = rand((V, H, 2), 4) * 0.5
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), 32, 32)
    out = torch.full((E, H), 7.0, device=DEV)
    gat.run_att(dev(att), out, 128, heads=H)
    ref = orc.gat_att(ptr, idx, att, H)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=RTOL)
    sums = np.add.reduceat(out.cpu().numpy().astype(np.float64), ptr[:-1][deg > 0], axis=0)
    np.testing.assert_allclose(sums, 1.0, rtol=1e-5)  # softmax rows sum to one
    if H == 1:
        a2 = att.reshape(V, 2)
        v = torch.empty(E, device=DEV)
        gat.run_u_add_v(dev(a2), v)
        assert np.array_equal(v.cpu().numpy(), orc.gat_u_add_v(ptr, idx, a2))
        w = torch.exp(torch.nn.functional.leaky_relu(v, 0.2))
        center = torch.full((V,), 7.0, device=DEV)
        gat.run_add_to_center(w, center)
        np.testing.assert_allclose(center.cpu().numpy(), orc.gat_add_to_center(ptr, w.cpu().numpy()), rtol=RTOL)
        assert np.all(center.cpu().numpy()[deg == 0] == 0)
        wh = w.cpu().numpy()
        gat.run_div_each(center, w)
        assert np.array_equal(w.cpu().numpy(), orc.gat_div_each(ptr, center.cpu().numpy(), wh))


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 5, 3), (300, 32, 128), (1000, 64, 64), (513, 100, 602), (129, 33, 7),
                                   # k_dense_nn_up<1..4> (K = 32 / 64 / 96 / 128: every chunk of a tile requested up front): ragged M, ragged N,
                                   # two column blocks, the arxiv-sized layer, 450 k rows (just under the row count where k_dense_nn_tall takes over)
                                   (169343, 32, 128), (169343, 64, 128), (100001, 33, 96), (70001, 7, 64), (50003, 64, 32), (127, 64, 128),
                                   (450000, 32, 128)])
def test_matmul_nn_bit_exact(M, N, K):
    """Dense combine GEMM (reference include/dense.h:4-23) on f32 MFMA: ascending-k fmaf chain == oracle."""
    A, B = rand((M, K), 1), rand((K, N), 2)
    C = gnc.matmul_NN(dev(A), dev(B))
    torch.cuda.synchronize()
    assert np.array_equal(C.cpu().numpy(), orc.matmul_nn(A, B))
    # asymmetric operands: a transposed C-write would not survive this
    np.testing.assert_allclose(C.cpu().numpy(), A.astype(np.float64) @ B.astype(np.float64), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(1500, 128, 512), (1100, 100, 70), (2049, 200, 33), (1024, 65, 1), (3000, 129, 602), (3000, 128, 602),
                                   (20001, 128, 36), (40003, 132, 34), (60001, 128, 33), (70536, 128, 40), (131072, 128, 8), (9000, 300, 64),
                                   (5000, 256, 602), (4099, 384, 128), (33000, 128, 30), (1025, 128, 2),
                                   # k_dense_nn_ahead / _ahead2 (chunks requested two periods ahead on fixed registers, hand-counted vmcnt): strips of
                                   # several tiles, one to sixty-four chunks per tile, ragged K, two column tiles
                                   (70536, 128, 64), (131072, 128, 96), (131072, 128, 32), (50001, 128, 70), (131072, 256, 34), (8191, 128, 2048),
                                   (300000, 128, 32), (60001, 128, 602)])
def test_matmul_nn_wide_kernel_bit_exact(M, N, K):
    """The wide-output kernels (taken for N > 64, M >= 1024 -- the 512 -> 128 layer): persistent strips of 32-row blocks walked in tiles
    of up to 128 rows, the chunk pipeline running across tiles.  k_dense_nn_lean (N % 128 == 0, A rows 16- or 8-byte aligned: buffer
    descriptors rebased per chunk, out-of-range rows / k zeroed by the hardware, masks on a ragged last K chunk only) and k_dense_nn_strip
    (everything else: ragged N, odd K, scalar loads).  Ragged M / N / K, every last-tile height (32 / 64 / 96 / 128 rows), strips of one
    and of several tiles, several column tiles, K smaller than a chunk -- still the oracle's ascending-k chain bit for bit.  Shapes with
    N % 128 == 0 and aligned rows run on k_dense_nn_ahead (K % 32 == 0, 16-byte rows) / k_dense_nn_ahead2 (even K, 8-byte rows)."""
    A, B = rand((M, K), 1), rand((K, N), 2)
    C = gnc.matmul_NN(dev(A), dev(B))
    torch.cuda.synchronize()
    assert np.array_equal(C.cpu().numpy(), orc.matmul_nn(A, B))
    np.testing.assert_allclose(C.cpu().numpy(), A.astype(np.float64) @ B.astype(np.float64), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("M,N,K", [(500001, 33, 100), (520000, 32, 128), (500017, 7, 68)])
def test_matmul_nn_tall_kernel_bit_exact(M, N, K):
    """The large-M kernel (k_dense_nn_tall: W held in registers, per-wavefront tiles, taken for M >= 500 k and 64 < K <= 128):
    same ascending-k chain as the oracle, ragged last tile / column block included."""
    A, B = rand((M, K), 5), rand((K, N), 6)
    C = gnc.matmul_NN(dev(A), dev(B))
    torch.cuda.synchronize()
    assert np.array_equal(C.cpu().numpy(), orc.matmul_nn(A, B))


def test_run_with_nn():
    V, E, F, OUT = 500, 9000, 128, 32
    ptr, idx = make_graph(V, E, seed=77)
    x, val, w = rand((V, F), 1), rand(E, 2), rand((F, OUT), 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, OUT)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [16])
    y = torch.full((V, F), 7.0, device=DEV)
    t = torch.full((V, OUT), 7.0, device=DEV)
    agg.run_with_nn(dev(x), y, dev(w), t, 128, 1)
    ps, tg = orc.neighbor_grouping(ptr, 16)
    y_ref = orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.mode_params("scheduled")[1])
    assert np.array_equal(y.cpu().numpy(), y_ref)
    assert np.array_equal(t.cpu().numpy(), orc.matmul_nn(y_ref, w))


@pytest.mark.parametrize("mode", ["rows", "scheduled", "balanced", "scheduled_items"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max"])
def test_gcn_fused_relu(mode, reduce):
    """GNNAGG_FLAG_RELU: y = max(A.x, 0) written by the aggregation kernels themselves -- short rows, segment rows, hubs
    folded in-kernel, the rows-mode long rows, and the item kernels + k_combine (NG = 2 schedule)."""
    V, E, F = 3000, 120000, 128
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=5, alpha=1.1)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    m = {"rows": 0, "scheduled": 1, "balanced": "balanced", "scheduled_items": 1}[mode]
    if mode == "scheduled":
        agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg.schedule(gnc.Schedule.neighbor_grouping, [32])
    if mode == "scheduled_items":
        agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg.schedule(gnc.Schedule.neighbor_grouping, [2])
    if mode == "balanced":
        agg.schedule_balanced(16)
    dx = dev(x)
    y_plain = torch.empty((V, F), device=DEV)
    agg.run(dx, y_plain, 128, m, reduce=reduce)
    y = torch.full((V, F), -7.0, device=DEV)
    agg.run(dx, y, 128, m, reduce=reduce, relu=True)
    assert torch.equal(y, torch.clamp_min(y_plain, 0.0))
    assert float(y_plain.min()) < 0.0


def test_gcn_fused_relu_with_accumulate():
    V, E, F = 2000, 50000, 64
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=6, alpha=1.1)
    ptr, idx = ptr_t.numpy().copy(), idx_t.numpy()
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.schedule_balanced(16)
    base = rand((V, F), 3)
    y_sum = torch.empty((V, F), device=DEV)
    agg.run(dev(x), y_sum, 128, "balanced")
    y = dev(base).clone()
    agg.run(dev(x), y, 128, "balanced", accumulate=True, relu=True)
    assert torch.equal(y, torch.clamp_min(dev(base) + y_sum, 0.0))


@pytest.mark.parametrize("F", [128, 48, 602])
def test_hub_fold_in_kernel_alternating_inputs(F):
    """Hubs are folded by the last segment workgroup to arrive, reading scratch rows other XCDs wrote (device-scope
    stores/loads, arrival counters reset in-kernel).  Alternate two inputs on one handle for many launches: a stale
    L2 line, a missed arrival or a counter left non-zero shows up as a mismatch against the oracle."""
    V, E = 6000, 400000
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=9, alpha=1.1)   # several rows with thousands of edges
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    val = rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.schedule_balanced(16)
    chunk, seg = agg.balanced_params()
    assert chunk == 16 and int(np.diff(ptr).max()) > 20 * chunk * seg     # hubs with > 16 segments exist
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    xs = [rand((V, F), 10 + i) for i in range(2)]
    refs = [orc.gcn_grouped(ps, tg, idx, val, x, V, seg=seg) for x in xs]
    dxs = [dev(x) for x in xs]
    y = torch.empty((V, F), device=DEV)
    for it in range(60):
        agg.run(dxs[it & 1], y, 128, "balanced")
        if it % 7 == 0 or it >= 56:
            assert np.array_equal(y.cpu().numpy(), refs[it & 1]), "launch %d" % it
    agg.run(dxs[0], y, 128, "balanced", reduce="max")          # a maximum does not depend on the association
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, xs[0]))
    agg.run(dxs[0], y, 128, "balanced", reduce="mean")         # the same fold, one IEEE division by the degree
    deg = np.maximum(np.diff(ptr), 1)[:, None].astype(np.float32)
    assert np.array_equal(y.cpu().numpy()[np.diff(ptr) > 0], (refs[0] / deg)[np.diff(ptr) > 0])


@pytest.mark.parametrize("mode", ["rows", "scheduled", "balanced"])
@pytest.mark.parametrize("F,OUT", [(128, 32), (32, 32), (64, 16), (100, 7), (256, 64), (30, 33), (602, 32), (7, 5)])
def test_run_with_nn_fused_epilogue(mode, F, OUT):
    """run_with_nn with the dense combine as the aggregation's epilogue (aggr_gcn.h:304-359,491-499): vout must equal
    the plain run and transformed must be BIT-exact the ascending-k chain of vout . W -- for short rows (fused tile),
    segment rows and hubs (row-list kernel), and the widths that fall back to the separate GEMM."""
    V, E = 3000, 60000
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=5, alpha=1.0)   # a few hub rows with many segments
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, val, w = rand((V, F), 1), rand(E, 2), rand((F, OUT), 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, OUT)
    m = {"rows": 0, "scheduled": 1, "balanced": "balanced"}[mode]
    if mode == "scheduled":
        agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg.schedule(gnc.Schedule.neighbor_grouping, [32])
    dx, dw = dev(x), dev(w)
    y_plain = torch.empty((V, F), device=DEV)
    agg.run(dx, y_plain, 128, m)
    y = torch.full((V, F), 7.0, device=DEV)
    t = torch.full((V, OUT), 7.0, device=DEV)
    agg.run_with_nn(dx, y, dw, t, 128, m)
    assert torch.equal(y, y_plain)
    assert np.array_equal(t.cpu().numpy(), orc.matmul_nn(y.cpu().numpy(), w))


def test_run_with_nn_mean_empty_rows():
    V, F, OUT = 70, 64, 32
    ptr = np.zeros(V + 1, np.int32)
    ptr[10:] = 3            # rows 0..8 empty, row 9 has 3 edges, the rest empty
    idx = np.array([1, 2, 3], np.int32)
    x, w = rand((V, F), 1), rand((F, OUT), 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, OUT)
    y = torch.full((V, F), 7.0, device=DEV)
    t = torch.full((V, OUT), 7.0, device=DEV)
    agg.run_with_nn(dev(x), y, dev(w), t, 128, "balanced")
    y_ref = orc.gcn_seq(ptr, idx, np.ones(3, np.float32), x)
    assert np.array_equal(y.cpu().numpy(), y_ref)
    assert np.array_equal(t.cpu().numpy(), orc.matmul_nn(y_ref, w))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("overlap,stages", [(False, 1), (True, 1), (True, ("stripe", 3)), (True, "owner")])
def test_partitioned_gcn_single_gpu_emulation(world, overlap, stages):
    """The row-partitioned plan of every rank, run one rank at a time on this GPU with the halo filled by hand
    (offline plan: the collective itself is covered by tests/test_dist_gloo.py).  stages: the staged exchange's plan -- the
    halo-source edges split by the stage their source arrives in, one accumulate pass per stage in stage order."""
    from gnn_computing_amd.dist import PartitionedGCN
    V, E, F = 4000, 90000, 128
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=11)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, val = rand((V, F), 1), rand(E, 2)
    y_ref = orc.gcn_seq(ptr, idx, val, x)
    scale = orc.gcn_abs_scale(ptr, idx, val, x)
    seen = 0
    for r in range(world):
        pg = PartitionedGCN(ptr, idx, val, F, device=DEV, rank=r, world=world, overlap=overlap, offline=True, stages=stages)
        hx = pg.hx
        assert hx.n_stages == (1 if stages == 1 else 3 if stages != "owner" else world - 1)
        r0, r1 = int(hx.bounds[r]), int(hx.bounds[r + 1])
        pg.set_local_x(dev(x[r0:r1]))
        pg.x_halo.copy_(dev(x[hx.halo_ids]))
        y = pg.step().cpu().numpy()
        seen += r1 - r0
        assert_within(y, y_ref[r0:r1], scale[r0:r1], "rank %d/%d" % (r, world))
        if overlap:
            # exact statement of the overlap plan: (balanced fold of local-source edges) + (balanced fold of halo-source edges)
            pl, il, pr, ir, is_loc = hx.split_local_remote()
            vl = val[hx.e0:hx.e1]
            cl, sl = pg.agg_loc.balanced_params()
            a = orc.gcn_grouped(*orc.neighbor_grouping(pl, cl), il, vl[is_loc], x[r0:r1], r1 - r0, seg=sl)
            deg = np.maximum(np.diff(hx.local_ptr), 1)[:, None].astype(np.float32)
            acc, acc_mean = a, a / deg
            for ag, (ps_, is_, m_) in zip(pg.agg_rem_stages, hx.split_remote_stages()):     # += stage 0, += stage 1, ... in order
                assert (ag is None) == (len(is_) == 0)
                if ag is None:
                    continue
                cr, sr = ag.balanced_params()
                b = orc.gcn_grouped(*orc.neighbor_grouping(ps_, cr), is_, vl[m_], x[hx.halo_ids], r1 - r0, seg=sr)
                has = (np.diff(ps_) > 0)[:, None]
                acc, acc_mean = np.where(has, acc + b, acc), np.where(has, acc_mean + b / deg, acc_mean)
            assert np.array_equal(y, acc)
            # mean and max on the overlap plan: the total degree divides every pass; a halo pass joins a maximum
            # only where the passes before it folded an edge (gnnagg_set_row_aux)
            ym = pg.step(reduce="mean").cpu().numpy()
            assert np.array_equal(ym, acc_mean)
            assert_within(ym, y_ref[r0:r1] / deg, scale[r0:r1] / deg, "mean, rank %d/%d" % (r, world))
            yx = pg.step(reduce="max").cpu().numpy()
            assert np.array_equal(yx, orc.gcn_max(ptr, idx, val, x)[r0:r1])
            ys = pg.step().cpu().numpy()                     # and back: the aux arrays do not leak into a sum
            assert np.array_equal(ys, y)
        else:
            ch, sg = pg.agg.balanced_params()
            ps, tg = orc.neighbor_grouping(hx.local_ptr, ch)
            x_ext = np.concatenate([x[r0:r1], x[hx.halo_ids]])
            assert np.array_equal(y, orc.gcn_grouped(ps, tg, hx.local_idx, val[hx.e0:hx.e1], x_ext, r1 - r0, seg=sg))
    assert seen == V


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("F,H", [(64, 1), (256, 8)])
@pytest.mark.parametrize("overlap,stages", [(False, 1), (True, 1), (True, ("stripe", 3)), (True, "owner")])
def test_partitioned_gat_single_gpu_emulation(world, F, H, overlap, stages):
    """PartitionedGAT (att rows travel with the feature rows): every rank's plan run on this GPU with the halo filled by
    hand, against the single-GPU fused result and against the rank-local order restated exactly.  overlap: numerators and
    denominators of the local-source edges first, the halo-source pass adds its own and divides (gnnagg_gat_run_part)."""
    from gnn_computing_amd.dist import PartitionedGAT
    V, E = 3000, 80000
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=17)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, att = rand((V, F), 1), rand((V, H, 2), 2) * 0.4
    y_ref = orc.gat_fused(ptr, idx, att, x, H)
    scale = gat_scale(ptr, idx, att, x, H) + np.abs(y_ref)
    seen = 0
    for r in range(world):
        pg = PartitionedGAT(ptr, idx, F, H, device=DEV, rank=r, world=world, offline=True, overlap=overlap, stages=stages)
        assert pg.overlap == overlap and pg.hx.n_stages == (1 if stages == 1 else 3 if stages != "owner" else world - 1)
        hx = pg.hx
        r0, r1 = int(hx.bounds[r]), int(hx.bounds[r + 1])
        n = r1 - r0
        x_ext = np.concatenate([x[r0:r1], x[hx.halo_ids]])
        att_ext = np.concatenate([att[r0:r1], att[hx.halo_ids]])
        pg.x_ext.copy_(dev(x_ext))
        pg.att_ext.copy_(dev(att_ext.reshape(len(att_ext), -1)))
        y = pg.compute().cpu().numpy()
        seen += n
        assert_within(y, y_ref[r0:r1], scale[r0:r1], "gat rank %d/%d" % (r, world))
        if overlap:
            # restated: (numerator, denominator) of the local-source edges + those of the halo-source edges, one division
            pl, il, pr, ir, _ = hx.split_local_remote()
            cl, sl = pg.agg_loc.balanced_params()
            _, _, (num, dn) = orc.gat_grouped(*orc.neighbor_grouping(pl, cl), il, att_ext, x_ext, n, H, seg=sl, parts=True)
            for ag, (ps_, is_, _) in zip(pg.agg_rem_stages, hx.split_remote_stages()):   # every stage's pass adds its own, the last divides
                if ag is None:
                    continue
                cr, sr = ag.balanced_params()
                _, _, (nb, db) = orc.gat_grouped(*orc.neighbor_grouping(ps_, cr), (is_ + n).astype(np.int32), att_ext, x_ext, n, H, seg=sr, parts=True)
                num, dn = num + nb, dn + db
            assert pg.agg_rem_stages[-1] is not None
            den = np.repeat(dn, F // H, axis=1)
            ref = np.where(den != 0, num / np.where(den != 0, den, 1), 0).astype(np.float32)
        else:
            ch, sg = pg.agg.balanced_params()
            assert pg.agg.balanced_partitions() == 0
            ref, _, _ = orc.gat_grouped(*orc.neighbor_grouping(hx.local_ptr, ch), hx.local_idx, att_ext, x_ext, n, H, seg=sg)
        assert_within(y, ref, scale[r0:r1], "gat rank %d/%d, restated order" % (r, world))
        assert np.all(y[np.diff(hx.local_ptr) == 0] == 0)
    assert seen == V


@pytest.mark.parametrize("F,H,chunk", [(128, 1, 64), (256, 8, 8), (96, 4, 5), (30, 3, 16)])
def test_gat_balanced_plan_with_hubs(F, H, chunk):
    """GAT balanced mode on the plan kernel: short rows, in-workgroup segment fold, hubs through scratch."""
    V = 350
    rng = np.random.default_rng(13)
    deg = rng.integers(0, 7, V)
    deg[11], deg[180], deg[349] = 4000, 900, 70
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    x, att = rand((V, F), 1), rand((V, H, 2), 2) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule_balanced(chunk)
    y = torch.full((V, F), 7.0, device=DEV)
    newval = torch.full((E, H), 7.0, device=DEV)
    gat.run(dev(x), dev(att), y, 128, "balanced", heads=H, newval=newval)
    ch, seg = gat.balanced_params()
    assert ch == chunk and seg == 16
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    ref, ref_newval, _ = orc.gat_grouped(ps, tg, idx, att, x, V, H, seg=seg)
    # same association as the oracle; only expf (device vs libm) differs by ulps
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=3e-6, atol=1e-6)
    np.testing.assert_allclose(newval.cpu().numpy(), ref_newval, rtol=1e-6)
    assert np.all(y.cpu().numpy()[deg == 0] == 0)


@pytest.mark.parametrize("F,H", [(128, 1), (256, 8), (96, 4)])
def test_gat_hub_fold_in_kernel_alternating_inputs(F, H):
    """GAT counterpart of test_hub_fold_in_kernel_alternating_inputs: numerators and per-head denominators of the hubs'
    segments cross XCDs through scratch; two inputs alternate on one handle."""
    V, E = 5000, 300000
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=19, alpha=1.1)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule_balanced(16)
    chunk, seg = gat.balanced_params()
    assert int(np.diff(ptr).max()) > 17 * chunk * seg
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    xs = [rand((V, F), 30 + i) for i in range(2)]
    atts = [rand((V, H, 2), 40 + i) * 0.4 for i in range(2)]
    refs = [orc.gat_grouped(ps, tg, idx, atts[i], xs[i], V, H, seg=seg)[0] for i in range(2)]
    dxs, datts = [dev(x) for x in xs], [dev(a) for a in atts]
    y = torch.empty((V, F), device=DEV)
    for it in range(40):
        gat.run(dxs[it & 1], datts[it & 1], y, 128, "balanced", heads=H)
        if it % 5 == 0 or it >= 37:
            np.testing.assert_allclose(y.cpu().numpy(), refs[it & 1], rtol=3e-6, atol=1e-6, err_msg="launch %d" % it)


@needs_extras
@pytest.mark.parametrize("F", [128, 32, 100, 33])
def test_gat_run_bwd(F):
    """run_bwd (aggr_gat.h:426-434): gradients of the single-head fused aggregation w.r.t. the input features and both
    attention terms, from the forward pass's newval / div -- hub rows on both the destination side (long rows) and the
    source side (popular nodes) -- against the double-precision oracle (itself checked by finite differences in
    tests/test_oracle_bwd.py)."""
    V, E = 3000, 90000
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=41, alpha=1.1)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, att, g = rand((V, F), 1), rand((V, 1, 2), 2) * 0.5, rand((V, F), 3)
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule_balanced(0)
    out = torch.empty((V, F), device=DEV)
    newval = torch.empty((E, 1), device=DEV)
    gat.run(dev(x), dev(att), out, 128, "balanced", heads=1, newval=newval)
    div = torch.empty(V, device=DEV)
    gat.run_add_to_center(newval, div)
    d_a_b = torch.full((V, 2), 7.0, device=DEV)
    d_feat = torch.full((V, F), 7.0, device=DEV)
    for _ in range(2):   # second call: cached transpose, same result
        gat.run_bwd(out, dev(g), newval, div, dev(x), d_a_b, d_feat, 0.2)
    ref_ab, ref_feat = orc.gat_bwd(ptr, idx, out.cpu().numpy(), g, newval.cpu().numpy(), div.cpu().numpy(), x, 0.2)
    scale_f = np.abs(ref_feat).max()
    scale_a = np.abs(ref_ab).max()
    np.testing.assert_allclose(d_feat.cpu().numpy(), ref_feat, rtol=1e-4, atol=1e-5 * scale_f)
    np.testing.assert_allclose(d_a_b.cpu().numpy(), ref_ab, rtol=1e-3, atol=2e-5 * scale_a)
    assert np.diff(ptr).max() > 1500                                  # destination hubs
    assert np.bincount(idx, minlength=V).max() > 500                  # source hubs


def test_balanced_mode_source_partitioned_on_high_degree_graph():
    """High average degree: the balanced mode picks the source-partitioned order (here forced to 16 column ranges, the reference's
    localityNeighborGrouping arrays, graph_schedule.h:156-243) -- GCN sum/mean/max, fused ReLU, accumulate (falls back to
    the chunked plan), run_with_nn and GAT, all against the oracle on the schedule the library reports."""
    V, E, F, H = 600, 200000, 64, 2
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=77, alpha=0.9)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("partitions", 16)     # (the library's own choice for a 600-column graph is one range: tests/test_gpu_blocked.py)
    assert agg.balanced_partitions() == 16
    chunk, seg = agg.balanced_params()
    assert seg == 0
    ps, ix, tg, vs = agg.get_schedule("balanced", with_val=True)
    ops, oix, otg, ovs = orc.locality_schedule(ptr, idx, 16, agg.balanced_partition_columns(), ng=chunk, val=val)
    assert np.array_equal(ps, ops) and np.array_equal(ix, oix) and np.array_equal(tg, otg) and np.array_equal(vs, ovs)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    ref = orc.gcn_grouped(ops, otg, oix, ovs, x, V, seg=0)
    assert np.array_equal(y.cpu().numpy(), ref)
    agg.run(dev(x), y, 128, "balanced", relu=True)
    assert np.array_equal(y.cpu().numpy(), np.maximum(ref, 0))
    deg = np.maximum(np.diff(ptr), 1)[:, None].astype(np.float32)
    agg.run(dev(x), y, 128, "balanced", reduce="mean")
    assert np.array_equal(y.cpu().numpy()[np.diff(ptr) > 0], (ref / deg)[np.diff(ptr) > 0])
    agg.run(dev(x), y, 128, "balanced", reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    # y += A.x: the chunked plan takes over for that call
    base = rand((V, F), 5)
    y.copy_(dev(base))
    agg.run(dev(x), y, 128, "balanced", accumulate=True)
    scale = orc.gcn_abs_scale(ptr, idx, val, x)
    assert np.all(np.abs(y.cpu().numpy() - (base + ref)) <= 1e-5 * (scale + np.abs(base)) + 1e-30)
    assert agg.balanced_partitions() == 16
    # dense combine behind it
    w = rand((F, 32), 6)
    t = torch.empty((V, 32), device=DEV)
    agg.run_with_nn(dev(x), y, dev(w), t, 128, "balanced")
    assert np.array_equal(y.cpu().numpy(), ref) and np.array_equal(t.cpu().numpy(), orc.matmul_nn(ref, w))
    # a non-square CSR (a rank's local graph indexes [X_local ; X_halo]): the ranges are cut from the real column count
    Vc = 1500
    idx2 = np.random.default_rng(3).integers(0, Vc, E).astype(np.int32)
    x2 = rand((Vc, F), 8)
    agg2 = gnc.Aggregator_GCN(dev(ptr), dev(idx2), dev(val), F, F)
    agg2.set_option("partitions", 16)
    assert agg2.balanced_partitions() == 16 and agg2.balanced_partition_columns() == int(idx2.max()) + 1
    agg2.run(dev(x2), y, 128, "balanced")
    s2 = orc.locality_schedule(ptr, idx2, 16, agg2.balanced_partition_columns(), ng=agg2.balanced_params()[0], val=val)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(s2[0], s2[2], s2[1], s2[3], x2, V, seg=0))
    # updateval aliasing (aggr_gcn.h:540-544) survives the library's permutation: values rewritten in place, then re-aliased
    dval = dev(val)
    agg3 = gnc.Aggregator_GCN(dev(ptr), dev(idx), dval, F, F)
    agg3.set_option("partitions", 16)
    agg3.run(dev(x), y, 128, "balanced")
    assert np.array_equal(y.cpu().numpy(), ref)
    val_b = rand(E, 12)
    dval.copy_(dev(val_b))
    agg3.run(dev(x), y, 128, "balanced")
    sb = orc.locality_schedule(ptr, idx, 16, agg3.balanced_partition_columns(), ng=chunk, val=val_b)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(sb[0], sb[2], sb[1], sb[3], x, V, seg=0))
    val_c = rand(E, 13)
    dval_c = dev(val_c)
    agg3.updateval(dval_c)
    agg3.run(dev(x), y, 128, "balanced")
    sc = orc.locality_schedule(ptr, idx, 16, agg3.balanced_partition_columns(), ng=chunk, val=val_c)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(sc[0], sc[2], sc[1], sc[3], x, V, seg=0))
    # the partitioned launch sequence (plan kernel + combine) captures into a hipGraph once the scratch is warm
    gr = torch.cuda.CUDAGraph()
    dx_in = dev(x)
    with torch.cuda.graph(gr):
        agg.run(dx_in, y, 128, "balanced")
    y.fill_(7.0)
    gr.replay()
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), ref)
    # an explicit chunk asks for the chunked order
    agg.schedule_balanced(64)
    assert agg.balanced_partitions() == 0 and agg.balanced_params() == (64, 16)
    # GAT
    att = rand((V, H, 2), 3) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("partitions", 16)
    assert gat.balanced_partitions() == 16
    yg = torch.full((V, F), 7.0, device=DEV)
    gat.run(dev(x), dev(att), yg, 128, "balanced", heads=H)
    gps, gix, gtg = gat.get_schedule("balanced")
    refg, _, _ = orc.gat_grouped(gps, gtg, gix, att, x, V, H, seg=0)
    np.testing.assert_allclose(yg.cpu().numpy(), refg, rtol=3e-6, atol=1e-6)


@needs_extras
@pytest.mark.parametrize("weights", [True, False])
def test_gcn_run_bwd_is_the_transposed_aggregation(weights):
    """d(input) = A^T d(output): checked through the adjoint identity <A x, g> == <x, A^T g> in float64 and against the
    float64 transposed product directly; hub rows on the source side included."""
    V, E, F = 4000, 120000, 96
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=51, alpha=1.1)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    val = rand(E, 2) if weights else None
    x, g = rand((V, F), 1), rand((V, F), 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if val is None else dev(val), F, F)
    dx = torch.full((V, F), 7.0, device=DEV)
    for _ in range(2):
        agg.run_bwd(dev(g), dx)
    rows = np.repeat(np.arange(V), np.diff(ptr))
    w = (val if val is not None else np.ones(E, np.float32)).astype(np.float64)
    ref = np.zeros((V, F))
    np.add.at(ref, idx, w[:, None] * g.astype(np.float64)[rows])
    scale = np.zeros((V, F))
    np.add.at(scale, idx, np.abs(w[:, None] * g.astype(np.float64)[rows]))
    assert np.all(np.abs(dx.cpu().numpy() - ref) <= 1e-5 * scale + 1e-30)
    y = torch.empty((V, F), device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    lhs = float((y.cpu().numpy().astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * dx.cpu().numpy()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float(np.abs(ref).sum())


@needs_extras
def test_autograd_wrappers_match_a_dense_torch_reference():
    """gcn_aggregate / gat_aggregate (HIP forward AND backward) against torch autograd on a dense fp32 restatement of the
    same layers -- a small graph with duplicate edges, empty rows and a hub."""
    V, F = 60, 24
    rng = np.random.default_rng(5)
    deg = rng.integers(0, 6, V)
    deg[7] = 200
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    val = rng.standard_normal(E).astype(np.float32)
    rows = torch.from_numpy(np.repeat(np.arange(V), deg)).to(DEV)
    cols = torch.from_numpy(idx.astype(np.int64)).to(DEV)
    g = dev(rand((V, F), 9))
    # GCN
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    x1 = dev(rand((V, F), 1)).requires_grad_(True)
    import gnn_computing_amd.extras  # noqa: F401  (raises without libgnnagg_extras.so)
    y1 = gnc.extras.gcn_aggregate(agg, x1)
    (y1 * g).sum().backward()
    x2 = x1.detach().clone().requires_grad_(True)
    A = torch.zeros((V, V), device=DEV).index_put_((rows, cols), dev(val), accumulate=True)
    y2 = A @ x2
    (y2 * g).sum().backward()
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-4) and torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4)
    # GAT, single head
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    xa = dev(rand((V, F), 2)).requires_grad_(True)
    aa = (dev(rand((V, 2), 3)) * 0.5).requires_grad_(True)
    ya = gnc.extras.gat_aggregate(gat, xa, aa)
    (ya * g).sum().backward()
    xb, ab = xa.detach().clone().requires_grad_(True), aa.detach().clone().requires_grad_(True)
    z = ab[rows, 0] + ab[cols, 1]
    w = torch.exp(torch.nn.functional.leaky_relu(z, 0.2))
    D = torch.zeros(V, device=DEV).index_add_(0, rows, w)
    yb = torch.zeros((V, F), device=DEV).index_add_(0, rows, (w / D[rows])[:, None] * xb[cols])
    (yb * g).sum().backward()
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-5)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(aa.grad, ab.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("chunk", [4, 64])
def test_plan_boundaries_exact_multiples(chunk):
    """Rows whose degree sits exactly on the plan's boundaries: chunk - 1, chunk, chunk + 1 (short row vs one segment),
    16 * chunk - 1, 16 * chunk, 16 * chunk + 1 (one segment vs hub with two), 32 * chunk, 17 * 16 * chunk + 3 (more than
    16 segments), next to empty rows -- GCN sum / max and GAT, bit-exact (GCN) against the oracle's restated order."""
    degs = [chunk - 1, chunk, chunk + 1, 0, 16 * chunk - 1, 16 * chunk, 16 * chunk + 1, 0, 0, 32 * chunk,
            17 * 16 * chunk + 3, 1, 2]
    V, F, H = len(degs), 72, 2
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(degs)
    E = int(ptr[-1])
    rng = np.random.default_rng(chunk)
    idx = rng.integers(0, V, E).astype(np.int32)
    x, val, att = rand((V, F), 1), rand(E, 2), rand((V, H, 2), 3) * 0.3
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.schedule_balanced(chunk)
    assert agg.balanced_params() == (chunk, 16) and agg.balanced_partitions() == 0
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=16))
    agg.run(dev(x), y, 128, "balanced", reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    agg.run(dev(x), y, 128, 0)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x))
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule_balanced(chunk)
    gat.run(dev(x), dev(att), y, 128, "balanced", heads=H)
    ref, _, _ = orc.gat_grouped(ps, tg, idx, att, x, V, H, seg=16)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=3e-6, atol=1e-6)


def test_run_clock_instrumentation():
    """run_clock (reference aggr_gcn.h:462-489, Figure 8): per-workgroup (start, end, CU id) stamps, results unchanged."""
    V, E, F = 3000, 40000, 64
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=21)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [16])
    y = torch.full((V, F), 7.0, device=DEV)
    hz = gnc.lib().gnnagg_wall_clock_hz()
    assert hz >= 1_000_000
    for sched, ref in ((0, orc.gcn_seq(ptr, idx, val, x)),
                       (1, orc.gcn_grouped(*orc.neighbor_grouping(ptr, 16), idx, val, x, V))):
        t = agg.run_clock(dev(x), y, 64, sched).cpu().numpy()
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), ref)
        busy = t[t[:, 1] != 0]
        assert len(busy) > 0.9 * len(t)
        assert np.all(busy[:, 1] >= busy[:, 0])                       # end after start
        span = (busy[:, 1].max() - busy[:, 0].min()) / hz
        assert 0 < span < 0.05                                         # the whole launch takes well under 50 ms
        assert len(np.unique(busy[:, 2])) > 8                          # workgroups ran on many CUs


@pytest.mark.parametrize("mode", ["balanced", "rows", "scheduled"])
def test_hipgraph_capture_and_replay(mode):
    """The launch-bound inner loop can be captured into a hipGraph (no allocation or synchronisation inside run()
    once the scratch is warm); the rows mode's auxiliary-stream fork/join is captured with it."""
    V, E, F = 6000, 120000, 128
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=31)
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    x1, x2, val = rand((V, F), 1), rand((V, F), 2), rand(E, 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule.neighbor_grouping, [16])
    dx, y = dev(x1), torch.empty((V, F), device=DEV)
    agg.run(dx, y, 512, mode)  # warm-up: plans, scratch
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        agg.run(dx, y, 512, mode)
    dx.copy_(dev(x2))  # new input, same buffers
    y.fill_(7.0)
    g.replay()
    torch.cuda.synchronize()
    if mode == "rows":
        ref = orc.gcn_seq(ptr, idx, val, x2)
    elif mode == "scheduled":
        ref = orc.gcn_grouped(*orc.neighbor_grouping(ptr, 16), idx, val, x2, V, seg=agg.mode_params("scheduled")[1])
    else:
        ch, sg = agg.balanced_params()
        ref = orc.gcn_grouped(*orc.neighbor_grouping(ptr, ch), idx, val, x2, V, seg=sg)
    assert np.array_equal(y.cpu().numpy(), ref)


@pytest.mark.parametrize("F,H", [(128, 1), (256, 8), (96, 4)])
def test_rows_mode_isolated_hub_gcn_and_gat(F, H):
    """`scheduled = 0` on a graph whose hub row holds a small share of the edges: the hub goes through the
    workgroup-per-row kernel on the auxiliary stream (GAT: when the head width is a multiple of 32), everything
    else through the descriptor path -- canonical CSR-order results either way."""
    V = 4000
    rng = np.random.default_rng(17)
    deg = rng.integers(0, 12, V)
    deg[123] = 3000  # ~12 % of the edges, far above 4x the average degree and above 1024
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    x, val, att = rand((V, F), 1), rand(E, 2), rand((V, H, 2), 3) * 0.4
    gcn = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    assert gcn.check_csr() == (0, 0)
    y = torch.full((V, F), 7.0, device=DEV)
    gcn.run(dev(x), y, 512, 0)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x))
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    y.fill_(7.0)
    gat.run(dev(x), dev(att), y, 128, 0, heads=H)
    ref = orc.gat_fused(ptr, idx, att, x, H)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=3e-6, atol=1e-6)
    assert np.all(y.cpu().numpy()[deg == 0] == 0)


@pytest.mark.parametrize("F", [30, 33, 130, 602, 64, 100, 128])
@pytest.mark.parametrize("weights", [True, False])
def test_rows_mode_long_rows_odd_widths_and_reductions(F, weights):
    """Rows mode with several long rows (1.5 k - 9 k edges: 4 - 21 rounds of the long-row kernel, ragged last round) at
    widths that take the 1- and 2-float lane packs and a ragged last 32-column tile, and at multiples of 4 (16-byte lanes: the form with
    two consumer wavefronts handing the accumulators to each other every 64 steps; F = 100: a last tile of 4 columns); sum / mean / max,
    bit-exact."""
    V = 3000
    rng = np.random.default_rng(23)
    deg = rng.integers(0, 10, V)
    deg[[5, 700, 1500, 2999]] = [9001, 1500, 4097, 2240]
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    x = rand((V, F), 1)
    val = rand(E, 2) if weights else None
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if val is None else dev(val), F, F)
    oval = val if val is not None else np.ones(E, np.float32)
    y = torch.full((V, F), 7.0, device=DEV)
    # "rows_hub_tile": 32-column tiles (one consumer lane per column in half a wavefront) or 64-column ones (the whole wavefront consumes:
    # the form a launch with many hub rows takes by itself); 64 with 1- and 2-float lanes has rounds of 56 / 112 edges (16-step batches)
    for tile_w in (32, 64):
        agg.set_option("rows_hub_tile", tile_w)
        for red, fn in (("sum", orc.gcn_seq), ("mean", orc.gcn_mean), ("max", orc.gcn_max)):
            y.fill_(7.0)
            agg.run(dev(x), y, 512, 0, reduce=red)
            assert np.array_equal(y.cpu().numpy(), fn(ptr, idx, oval, x)), (red, tile_w)


@pytest.mark.parametrize("F,H", [(128, 4), (30, 1), (33, 1), (602, 1), (256, 8), (96, 4)])
@pytest.mark.parametrize("medium", [0, 16, 700, -1])
def test_rows_mode_medium_rows_take_the_128_thread_form(F, H, medium):
    """Rows between the lane-group class and the hub class (option "rows_medium_edges": 0 = the library's rule -- 128 edges on a
    graph this small --, -1 = no such class) run the long-row kernel with ONE gather wavefront: 64 edges per round, so rows of
    130 ... 1000 edges take 3 ... 16 rounds with ragged last ones, beside hub rows on the 512-thread form and short rows on lane
    groups.  The chain is the reference's sequential one whichever class a row falls in: sum / mean / max bit-exact for every
    threshold, GAT (head width a multiple of 32) within the fused bound, and the fused dense combine sees the medium rows' products."""
    V = 3000
    rng = np.random.default_rng(29)
    deg = rng.integers(0, 10, V)
    deg[[3, 40, 41, 900, 1700, 2500, 2998]] = [130, 191, 257, 640, 1000, 1023, 5000]
    deg[rng.choice(np.arange(100, 800), 40, replace=False)] = rng.integers(129, 400, 40)
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    idx = rng.integers(0, V, E).astype(np.int32)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("rows_medium_edges", medium)
    y = torch.full((V, F), 7.0, device=DEV)
    for red, fn in (("sum", orc.gcn_seq), ("mean", orc.gcn_mean), ("max", orc.gcn_max)):
        y.fill_(7.0)
        agg.run(dev(x), y, 512, 0, reduce=red)
        assert np.array_equal(y.cpu().numpy(), fn(ptr, idx, val, x)), red
    if F == 128:   # aggregation -> dense combine in one call: the medium rows' products come from the rows-list GEMM
        w = rand((F, 48), 5)
        out = torch.full((V, 48), 7.0, device=DEV)
        agg.run_with_nn(dev(x), y, dev(w), out, 512, 0)
        ysum = orc.gcn_seq(ptr, idx, val, x)
        assert np.array_equal(y.cpu().numpy(), ysum)
        assert np.array_equal(out.cpu().numpy(), orc.matmul_nn(ysum, w))
    if H > 1:   # (head width 24: a 32-column tile straddles heads -- the long-row forms do not apply, the row kernels take every row)
        att = rand((V, H, 2), 3) * 0.4
        gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        gat.set_option("rows_medium_edges", medium)
        y.fill_(7.0)
        gat.run(dev(x), dev(att), y, 128, 0, heads=H)
        np.testing.assert_allclose(y.cpu().numpy(), orc.gat_fused(ptr, idx, att, x, H), rtol=3e-6, atol=1e-6)


def test_check_csr_flags_bad_input():
    ptr = np.array([0, 2, 1, 4], np.int32)           # row 1 has ptr[1] > ptr[2]
    idx = np.array([0, 5, 2, -1], np.int32)          # 5 and -1 are outside [0, 3)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, 4, 4)
    assert agg.check_csr() == (1, 2)
    assert agg.check_csr(num_cols=6) == (1, 1)


def test_fuzz_gcn_modes_reductions_alignment():
    """Randomised sweep: graph shape, feature width (incl. odd widths), mode, reduction, weights on/off and
    deliberately mis-aligned feature buffers (forces the narrower vector paths) -- always against the oracle."""
    import os
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2024")))
    for case in range(int(os.environ.get("FUZZ_CASES", "60"))):
        V = int(rng.integers(1, 400))
        E = int(rng.integers(0, 30 * V))
        F = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 128, 130, 256]))
        ptr, idx = gnc.graph.uniform_random_csr(V, E, seed=1000 + case)
        if E and rng.random() < 0.3:  # a hub row
            hub = int(rng.integers(0, V))
            deg = np.diff(ptr).astype(np.int64)
            deg[hub] += int(rng.integers(200, 3000))
            ptr = np.zeros(V + 1, np.int32)
            ptr[1:] = np.cumsum(deg)
            idx = rng.integers(0, V, int(ptr[-1])).astype(np.int32)
            E = int(ptr[-1])
        x = rng.standard_normal((V, F)).astype(np.float32)
        val = rng.standard_normal(E).astype(np.float32) if rng.random() < 0.7 else None
        off = int(rng.choice([0, 0, 1, 2]))  # element offset of the feature buffers inside their allocation
        xb = torch.zeros(V * F + 4, device=DEV)
        yb = torch.full((V * F + 4,), 7.0, device=DEV)
        dx = xb[off:off + V * F].view(V, F)
        dy = yb[off:off + V * F].view(V, F)
        dx.copy_(dev(x))
        agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if val is None else dev(val), F, F)
        mode = str(rng.choice(["rows", "scheduled", "balanced"]))
        reduce = str(rng.choice(["sum", "mean", "max"]))
        ng = int(rng.choice([1, 3, 16, 32, 64]))
        if mode == "scheduled":
            agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
            agg.schedule(gnc.Schedule.neighbor_grouping, [ng])
        elif mode == "balanced":
            agg.schedule_balanced(int(rng.choice([0, 4, 64])))
        agg.run(dx, dy, 512, mode, reduce=reduce)
        got = dy.cpu().numpy()
        what = "case %d V=%d E=%d F=%d %s %s off=%d" % (case, V, E, F, mode, reduce, off)
        if reduce == "max":
            assert np.array_equal(got, orc.gcn_max(ptr, idx, val, x)), what
            continue
        if mode == "rows":
            ref = orc.gcn_seq(ptr, idx, val, x) if reduce == "sum" else orc.gcn_mean(ptr, idx, val, x)
            assert np.array_equal(got, ref), what
            continue
        if mode == "scheduled":
            ps, tg = orc.neighbor_grouping(ptr, ng)
            ref = orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.mode_params("scheduled")[1])
        elif agg.balanced_partitions() > 0:   # tiny V, hundreds of edges per row: the source-partitioned order
            ch, sg = agg.balanced_params()
            ps, ix, tg, vs = orc.locality_schedule(ptr, idx, agg.balanced_partitions(), agg.balanced_partition_columns(),
                                                   ng=ch, val=val)
            ref = orc.gcn_grouped(ps, tg, ix, vs, x, V, seg=0)
        else:
            ch, sg = agg.balanced_params()
            ps, tg = orc.neighbor_grouping(ptr, ch)
            ref = orc.gcn_grouped(ps, tg, idx, val, x, V, seg=sg)
        if reduce == "mean":
            ref = ref / np.maximum(np.diff(ptr), 1)[:, None].astype(np.float32)
        assert np.array_equal(got, ref), what
        assert float(yb[:off].sum()) == 7.0 * off and float(yb[off + V * F:].sum()) == 7.0 * (4 - off), what + " (out of bounds write)"


def test_fuzz_gat_modes_heads():
    """Randomised GAT sweep: heads, head width (incl. widths that force the scalar path), mode, hub rows."""
    import os
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "4242")))
    for case in range(int(os.environ.get("FUZZ_CASES", "40"))):
        V = int(rng.integers(2, 300))
        E = int(rng.integers(1, 20 * V))
        H = int(rng.choice([1, 2, 3, 4, 8]))
        D = int(rng.choice([1, 2, 4, 5, 8, 16, 32, 64]))
        F = H * D
        ptr, idx = gnc.graph.uniform_random_csr(V, E, seed=3000 + case)
        if rng.random() < 0.3:
            deg = np.diff(ptr).astype(np.int64)
            deg[int(rng.integers(0, V))] += int(rng.integers(300, 2500))
            ptr = np.zeros(V + 1, np.int32)
            ptr[1:] = np.cumsum(deg)
            idx = rng.integers(0, V, int(ptr[-1])).astype(np.int32)
            E = int(ptr[-1])
        x = rng.standard_normal((V, F)).astype(np.float32)
        att = (rng.standard_normal((V, H, 2)) * 0.5).astype(np.float32)
        gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        mode = str(rng.choice(["rows", "scheduled", "balanced"]))
        ng = int(rng.choice([2, 16, 32]))
        if mode == "scheduled":
            gat.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
            gat.schedule(gnc.Schedule.neighbor_grouping, [ng])
        elif mode == "balanced":
            gat.schedule_balanced(int(rng.choice([0, 4, 64])))
        y = torch.full((V, F), 7.0, device=DEV)
        gat.run(dev(x), dev(att), y, 128, mode, heads=H)
        if mode == "rows":
            ref = orc.gat_fused(ptr, idx, att, x, H)
        elif mode == "scheduled":
            ref = orc.gat_grouped(*orc.neighbor_grouping(ptr, ng), idx, att, x, V, H, seg=gat.mode_params("scheduled")[1])[0]
        else:
            ch, sg = gat.balanced_params()
            ref = orc.gat_grouped(*orc.neighbor_grouping(ptr, ch), idx, att, x, V, H, seg=sg)[0]
        what = "case %d V=%d E=%d H=%d D=%d %s" % (case, V, E, H, D, mode)
        # same association as the kernel; device expf vs libm differ by ulps per edge: 1e-5 of the weighted magnitude
        assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), what)
        assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0), what


@pytest.mark.parametrize("F", [1, 4, 33, 128])
def test_degenerate_graphs(F):
    """Graphs at the edge of the input domain: no edges at all (every row empty: Y = 0 in every mode, the reference memsets vout,
    aggr_gcn.h:393), a single row with a self loop, a single row with many parallel edges to itself, rows that all name one source;
    GCN in the three modes and all reductions, GAT fused, the edge-softmax stages, schedules on them."""
    cases = []
    V = 5
    cases.append((np.zeros(V + 1, np.int32), np.zeros(0, np.int32)))                                   # edgeless
    cases.append((np.array([0, 1], np.int32), np.array([0], np.int32)))                                # one self loop
    cases.append((np.array([0, 300], np.int32), np.zeros(300, np.int32)))                              # 300 parallel self loops
    cases.append((np.arange(0, 4 * 7 + 1, 7, dtype=np.int32), np.full(28, 2, np.int32)))               # 4 rows x 7 edges, all from row 2
    for ptr, idx in cases:
        V, E = len(ptr) - 1, len(idx)
        x, val = rand((V, F), 1), rand(E, 2)
        agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val) if E else None, F, F)
        agg.set_option("fast_scheduled", 0)
        agg.schedule(gnc.Schedule.neighbor_grouping, [4])
        v_or_none = val if E else None
        for mode in (0, 1, "balanced"):
            for red, ref in (("sum", orc.gcn_seq), ("mean", orc.gcn_mean), ("max", orc.gcn_max)):
                y = torch.full((V, F), 7.0, device=DEV)
                agg.run(dev(x), y, 128, mode, reduce=red)
                expect = ref(ptr, idx, v_or_none if E else np.zeros(0, np.float32), x)
                scale = orc.gcn_abs_scale(ptr, idx, v_or_none if E else np.zeros(0, np.float32), x)
                assert_within(y.cpu().numpy(), expect, scale + np.abs(expect), "degenerate %s %s V=%d E=%d" % (mode, red, V, E))
                if E == 0:
                    assert torch.count_nonzero(y) == 0
        H = 1
        att = rand((V, H, 2), 3) * 0.4
        gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        for mode in (0, "balanced"):
            y = torch.full((V, F), 7.0, device=DEV)
            gat.run(dev(x), dev(att), y, 128, mode, heads=H)
            expect = orc.gat_fused(ptr, idx, att, x, H)
            assert_within(y.cpu().numpy(), expect, np.abs(expect) + 1e-6, "degenerate gat %s V=%d E=%d" % (mode, V, E))
        out = torch.full((max(E, 1), H), 7.0, device=DEV)[:E]
        gat.run_att(dev(att), out, 128, heads=H)
        if E:
            np.testing.assert_allclose(out.cpu().numpy(), orc.gat_att(ptr, idx, att, H), rtol=RTOL)
