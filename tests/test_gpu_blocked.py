"""GPU parity of the round-2 additions around the balanced mode: the 2-D blocked (source range x column tile) order on the
column-tiled image of X, handle options, the gather probe, permuted schedules following updateval, run_clock's capacity
check.  Same bar as tests/test_gpu_parity.py: bit-exact wherever the kernel keeps the oracle's association."""
import ctypes

import numpy as np
import pytest
import torch

import gnn_computing_amd as gnc
from gnn_computing_amd import _lib
from oracle import oracle as orc
from test_gpu_parity import DEV, assert_within, dev, gat_scale, rand

pytestmark = pytest.mark.gpu

# the older kernel forms ("retile", "tiled", "spans", "inkernel_combine", "host_plan") are compile-time constants in the shipped library and
# run-time options of libgnnagg_extras.so only: their A / B cases run in the second tier
# (GNNAGG_LIB=gnn_computing_amd/libgnnagg_extras.so GNNAGG_TEST_TIER=2), the default-form cases in every pass
from gnn_computing_amd import _lib as _gl
LEGACY = {"retile", "tiled", "spans", "inkernel_combine", "host_plan", "partition_min_degree"}


def shipped(opts):
    """True when this option set can be applied to the library under test"""
    return _gl.has_extras() or not (set(opts) & LEGACY)


def hub_graph(V, E, seed, alpha=0.9):
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=seed, alpha=alpha)
    return ptr_t.numpy(), idx_t.numpy()


def blocked_reference(agg, ptr, idx, val):
    """The order the library reports, restated with the oracle's scheduler: groups of localityNeighborGrouping
    (graph_schedule.h:156-243) for (partitions, chunk, column count), folded flat in ascending group order."""
    parts, cols = agg.balanced_partitions(), agg.balanced_partition_columns()
    chunk, seg = agg.balanced_params()
    assert parts >= 1 and seg == 0
    return orc.locality_schedule(ptr, idx, parts, cols, ng=chunk, val=val)


@pytest.mark.parametrize("F", [602, 100, 64, 30, 33, 256, 8])
@pytest.mark.parametrize("slice_kb,tile_w", [(16, 64), (64, 32), (4, 128)])
def test_blocked_gcn_matches_the_restated_order(F, slice_kb, tile_w):
    """Auto-chosen range count from (columns x tile bytes / slice), every column-tile geometry, Y rows that are 16-, 8- and
    4-byte aligned (F = 602 is BASELINE's SAGE width: 8-byte rows, ragged last tile), sum / mean / max / fused ReLU."""
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=5)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("slice_kb", slice_kb)
    agg.set_option("tile_width", tile_w)
    parts, cols = agg.balanced_partitions(), agg.balanced_partition_columns()
    assert cols == int(idx.max()) + 1
    expect = min(-(-cols * tile_w * 4 // (slice_kb * 1024)), max(1, (E // V) // 12))   # slice-sized, >= 12 edges per sub-row
    assert parts == expect and parts > 1
    ps, ix, tg, vs = blocked_reference(agg, ptr, idx, val)
    got = agg.get_schedule("balanced", with_val=True)
    assert all(np.array_equal(a, b) for a, b in zip(got, (ps, ix, tg, vs)))
    ref = orc.gcn_grouped(ps, tg, ix, vs, x, V, seg=0)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    assert np.array_equal(y.cpu().numpy(), ref)
    agg.run(dev(x), y, 128, "balanced", relu=True)
    assert np.array_equal(y.cpu().numpy(), np.maximum(ref, 0))
    deg = np.diff(ptr)
    agg.run(dev(x), y, 128, "balanced", reduce="mean")
    assert np.array_equal(y.cpu().numpy()[deg > 0], (ref / np.maximum(deg, 1)[:, None].astype(np.float32))[deg > 0])
    assert np.all(y.cpu().numpy()[deg == 0] == 0)
    agg.run(dev(x), y, 128, "balanced", reduce="max")
    assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x))
    # within the fp32 bound of the canonical CSR-order chain (north_star's 1e-5)
    agg.run(dev(x), y, 128, "balanced")
    assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x), "blocked vs CSR order")
    # implicit unit weights and an x / y pair that is only 4-byte aligned
    agg1 = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    agg1.set_option("slice_kb", slice_kb)
    agg1.set_option("tile_width", tile_w)
    xb = torch.empty(V * F + 1, device=DEV)
    yb = torch.full((V * F + 1,), 7.0, device=DEV)
    xb[1:].copy_(dev(x).reshape(-1))
    agg1.run_with_feat(xb[1:].view(V, F), yb[1:].view(V, F), 128, "balanced", F)
    s1 = blocked_reference(agg1, ptr, idx, None)
    assert np.array_equal(yb[1:].view(V, F).cpu().numpy(), orc.gcn_grouped(s1[0], s1[2], s1[1], None, x, V, seg=0))
    assert float(yb[0]) == 7.0


def test_blocked_mode_a_b_switches_give_identical_results():
    """The segmented-stream kernel, one descriptor per lane group, tile-major on the tiled image or on the caller's X
    (line-aligned rows), and the round-1 order without column tiles differ in schedule only: the same bits."""
    V, E, F = 700, 220000, 256
    ptr, idx = hub_graph(V, E, seed=6)
    x, val = rand((V, F), 3), rand(E, 4)
    for opts in ({}, {"retile": 0}, {"tiled": 0}, {"tile_width": 256}, {"spans": 0}, {"spans": 0, "retile": 0}, {"tile_width": 32}):
        if not shipped(opts):
            continue
        agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
        agg.set_option("partitions", 8)
        for k, v in opts.items():
            agg.set_option(k, v)
        y = torch.full((V, F), 7.0, device=DEV)
        agg.run(dev(x), y, 128, "balanced")
        chunk = agg.balanced_params()[0]          # (spans cut groups at 128 edges, the descriptor kernels at pick_chunk)
        ps, ix, tg, vs = orc.locality_schedule(ptr, idx, 8, int(idx.max()) + 1, ng=chunk, val=val)
        assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, vs, x, V, seg=0)), str(opts)


@pytest.mark.parametrize("F,H", [(256, 8), (64, 1), (96, 4), (30, 3), (32, 1), (16, 2)])
def test_blocked_gat_and_newval_in_csr_edge_order(F, H):
    """GAT on the blocked order (head width 3 is not a multiple of the 16-byte lanes: the library keeps the row-major
    geometry there), and the un-normalised weights come back in CSR edge order although the kernel walks a permuted
    edge list (round-1 advisor finding)."""
    V, E = 800, 200000
    ptr, idx = hub_graph(V, E, seed=7)
    x, att = rand((V, F), 1), rand((V, H, 2), 2) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("slice_kb", 16)
    parts = gat.balanced_partitions()
    assert parts > 1
    y = torch.full((V, F), 7.0, device=DEV)
    newval = torch.full((E, H), 7.0, device=DEV)
    gat.run(dev(x), dev(att), y, 128, "balanced", heads=H, newval=newval)
    ps, ix, tg = gat.get_schedule("balanced")
    ops, oix, otg, _ = orc.locality_schedule(ptr, idx, parts, gat.balanced_partition_columns(), ng=gat.balanced_params()[0])
    assert np.array_equal(ps, ops) and np.array_equal(ix, oix) and np.array_equal(tg, otg)
    ref, _, _ = orc.gat_grouped(ops, otg, oix, att, x, V, H, seg=0)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "blocked gat")
    assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0)
    # newval: CSR edge order, the same weights the canonical rows mode writes
    _, ref_newval, _ = orc.gat_grouped(*orc.neighbor_grouping(ptr, 1 << 30), idx, att, x, V, H, seg=0)
    np.testing.assert_allclose(newval.cpu().numpy(), ref_newval, rtol=1e-6)
    nv_rows = torch.full((E, H), 7.0, device=DEV)
    gat.run(dev(x), dev(att), y, 128, "rows", heads=H, newval=nv_rows)
    assert torch.equal(newval, nv_rows)


@pytest.mark.parametrize("F,H,opts", [(96, 3, {"retile": 0}), (96, 3, {}), (256, 8, {"retile": 0}), (256, 8, {"tile_width": 128}),
                                      (256, 8, {"tile_width": 256}), (128, 4, {"tile_width": 32}), (64, 1, {"tile_width": 32}),
                                      (256, 2, {"tile_width": 64})])
def test_blocked_gat_span_kernel_variants(F, H, opts):
    """k_gat_span's instantiations: heads per tile 1 / 2 / 4 / 8 (the compact attention image replicates the last head where a
    tile reaches beyond it: 3 heads of 32 on 64-float tiles), a head wider than the tile, gathers from the caller's X with a
    row pitch that is not a power of two (96 floats: the 64-bit address path) and from the tiled image (shift-or offsets)."""
    if not shipped(opts):
        pytest.skip("second tier: an older kernel form, libgnnagg_extras.so only")
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=11)
    x, att = rand((V, F), 5), rand((V, H, 2), 6) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("slice_kb", 16)
    for k, v in opts.items():
        gat.set_option(k, v)
    parts = gat.balanced_partitions()
    assert parts > 1
    y = torch.full((V, F), 7.0, device=DEV)
    newval = torch.full((E, H), 7.0, device=DEV)
    gat.run(dev(x), dev(att), y, 128, "balanced", heads=H, newval=newval)
    ops, oix, otg, _ = orc.locality_schedule(ptr, idx, parts, gat.balanced_partition_columns(), ng=gat.balanced_params()[0])
    ref, _, _ = orc.gat_grouped(ops, otg, oix, att, x, V, H, seg=0)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "blocked gat %s" % opts)
    assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0)
    _, ref_newval, _ = orc.gat_grouped(*orc.neighbor_grouping(ptr, 1 << 30), idx, att, x, V, H, seg=0)
    np.testing.assert_allclose(newval.cpu().numpy(), ref_newval, rtol=1e-6)


def test_single_range_is_the_library_choice_for_small_graphs():
    """600 columns x 256 B fit any L2: one range, tile-major order only; rows longer than the chunk still split."""
    V, E, F = 600, 200000, 128
    ptr, idx = hub_graph(V, E, seed=8)
    x = rand((V, F), 1)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    assert agg.balanced_partitions() == 1
    y = torch.empty((V, F), device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    ps, ix, tg, _ = blocked_reference(agg, ptr, idx, None)
    assert np.array_equal(ix, idx) and int(np.diff(ps).max()) <= agg.balanced_params()[0] < int(np.diff(ptr).max())
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, None, x, V, seg=0))
    agg.set_option("partitions", 0)      # never partition: the chunked plan with its 16-chunk segment fold
    assert agg.balanced_partitions() == 0 and agg.balanced_params()[1] == 16
    agg.run(dev(x), y, 128, "balanced")
    chunk, seg = agg.balanced_params()
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(*orc.neighbor_grouping(ptr, chunk), idx, None, x, V, seg=seg))


def test_fast_rows_option_gives_scheduled0_the_balanced_order():
    """reference drivers call run(vin, vout, B, 0) (aggr_gcn.h:379-410); with the option that call runs the balanced order:
    within 1e-5 of the canonical chain instead of bit-equal to it, and restatable."""
    V, E, F = 5000, 150000, 128
    ptr, idx = hub_graph(V, E, seed=9, alpha=1.0)
    x, val = rand((V, F), 1), rand(E, 2)
    seq = orc.gcn_seq(ptr, idx, val, x)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    y = torch.empty((V, F), device=DEV)
    agg.run(dev(x), y, 512, 0)
    assert np.array_equal(y.cpu().numpy(), seq)                      # default: canonical CSR-order chains
    agg.set_option("fast_rows", 1)
    agg.run(dev(x), y, 512, 0)
    chunk, seg = agg.mode_params("rows")
    assert (chunk, seg) == agg.balanced_params() and seg == 16
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(*orc.neighbor_grouping(ptr, chunk), idx, val, x, V, seg=seg))
    assert_within(y.cpu().numpy(), seq, orc.gcn_abs_scale(ptr, idx, val, x), "fast rows vs CSR order")
    agg.set_option("fast_rows", 0)
    agg.run(dev(x), y, 512, 0)
    assert np.array_equal(y.cpu().numpy(), seq)
    # GAT
    att = rand((V, 2), 3) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("fast_rows", 1)
    gat.run(dev(x), dev(att), y, 128, 0)
    ch, sg = gat.balanced_params()
    ref, _, _ = orc.gat_grouped(*orc.neighbor_grouping(ptr, ch), idx, att, x, V, 1, seg=sg)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "fast rows gat")


def test_aux_stream_option_keeps_rows_mode_bit_exact_on_one_stream():
    """rows mode with hub rows: "aux_stream" = 0 runs the long-row kernel on the handle's stream instead of an auxiliary one."""
    V, E, F = 5000, 150000, 128
    ptr, idx = hub_graph(V, E, seed=9, alpha=1.0)
    x, val, att = rand((V, F), 1), rand(E, 2), rand((V, 2), 3) * 0.4
    seq = orc.gcn_seq(ptr, idx, val, x)
    for aux in (0, 1):
        agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
        agg.set_option("aux_stream", aux)
        y = torch.full((V, F), 7.0, device=DEV)
        agg.run(dev(x), y, 512, 0)
        assert np.array_equal(y.cpu().numpy(), seq), aux
        gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        gat.set_option("aux_stream", aux)
        gat.run(dev(x), dev(att), y, 128, 0)
        ref = orc.gat_fused(ptr, idx, att, x)
        assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "gat rows aux=%d" % aux)


def test_scheduled1_runs_the_balanced_order_by_default_and_the_users_groups_on_request():
    """reference drivers call schedule(neighbor_grouping, {NG}) + run(vin, vout, B, 1).  Default ("fast_scheduled" = 1): that call
    runs the balanced order (same bits as mode "balanced": the reference's aggr_gcn_target adds with atomicAdd, aggr_gcn.h:112, so
    any association is one of its results), the user's groups keep describing num_target / get_schedule, and a scheduled run
    without a schedule still fails.  "fast_scheduled" = 0: the user's groups in the restated order, bit-exact."""
    V, E, F = 5000, 150000, 128
    ptr, idx = hub_graph(V, E, seed=9, alpha=1.0)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    y = torch.empty((V, F), device=DEV)
    with pytest.raises(Exception):
        agg.run(dev(x), y, 512, 1)
    agg.schedule(gnc.Schedule.neighbor_grouping, [32])
    ps, tg = orc.neighbor_grouping(ptr, 32)
    agg.run(dev(x), y, 512, 1)
    yb = torch.empty((V, F), device=DEV)
    agg.run(dev(x), yb, 512, "balanced")
    assert torch.equal(y, yb)
    chunk, seg = agg.balanced_params()
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(*orc.neighbor_grouping(ptr, chunk), idx, val, x, V, seg=seg))
    assert agg.num_target == len(tg) and np.array_equal(agg.get_schedule("scheduled")[0], ps)
    assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x), "fast scheduled vs CSR order")
    agg.set_option("fast_scheduled", 0)
    agg.run(dev(x), y, 512, 1)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=agg.mode_params("scheduled")[1]))
    agg.set_option("fast_scheduled", 1)
    agg.run(dev(x), y, 512, 1)
    assert torch.equal(y, yb)
    att = rand((V, 2), 3) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule(gnc.Schedule.neighbor_grouping, [32])
    gat.run(dev(x), dev(att), y, 128, 1)
    gat.run(dev(x), dev(att), yb, 128, "balanced")
    assert torch.equal(y, yb)
    # a GAT call that asks for newval keeps the user's groups (newval is defined per scheduled edge weight)
    nv = torch.empty(E, device=DEV)
    gat.run(dev(x), dev(att), y, 128, 1, newval=nv)
    ref, ref_nv, _ = orc.gat_grouped(ps, tg, idx, att, x, V, 1, seg=gat.mode_params("scheduled")[1])
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "gat scheduled with newval")


def test_reference_facing_surfaces_default_to_the_balanced_order_for_scheduled0():
    """The pybind-named functions (Figure7/kernel.cpp:166-179) make handles with the reference-facing defaults: gcn_run(., ., ., B, 0)
    runs the balanced order (within 1e-5 of aggr_gcn's chain, bit-equal to mode "balanced"); the class API of this package keeps
    the canonical chains for scheduled = 0; GNNAGG_MODE_ROWS on a section-B handle stays canonical."""
    V, E, F = 5000, 150000, 128
    ptr, idx = hub_graph(V, E, seed=9, alpha=1.0)
    x, val = rand((V, F), 1), rand(E, 2)
    seq, scale = orc.gcn_seq(ptr, idx, val, x), orc.gcn_abs_scale(ptr, idx, val, x)
    at = gnc.gcn_init(dev(ptr), dev(idx), dev(val))
    y, yb = torch.empty((V, F), device=DEV), torch.empty((V, F), device=DEV)
    gnc.gcn_run(at, dev(x), y, 128, 0)
    at.run(dev(x), yb, 128, "balanced")
    assert torch.equal(y, yb)
    assert_within(y.cpu().numpy(), seq, scale, "reference-facing scheduled = 0")
    at.set_option("fast_rows", 0)
    gnc.gcn_run(at, dev(x), y, 128, 0)
    assert np.array_equal(y.cpu().numpy(), seq)
    cls = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    cls.run(dev(x), y, 512, 0)
    assert np.array_equal(y.cpu().numpy(), seq)
    att = rand((V, 2), 3) * 0.4
    g = gnc.gat_init(dev(ptr), dev(idx))
    gnc.gat_run(g, dev(x), dev(att), y, 128, 0)
    g.run(dev(x), dev(att), yb, 128, "balanced")
    assert torch.equal(y, yb)


@pytest.mark.parametrize("kind,param", [("locality", [3]), ("locality_neighbor_grouping", [4, 8])])
def test_user_locality_schedule_follows_updateval(kind, param):
    """A locality schedule permutes the edge values into its own copy (aggr_gcn.h:509-537); updateval (:540-544) and
    in-place rewrites of the caller's array must reach it (round-1 advisor finding)."""
    V, E, F = 200, 6000, 64
    ptr, idx = gnc.graph.uniform_random_csr(V, E, seed=12)
    x, v1, v2, v3 = rand((V, F), 1), rand(E, 2), rand(E, 3), rand(E, 4)
    dv = dev(v1)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dv, F, F)
    agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
    agg.schedule(gnc.Schedule[kind], param)
    ng = param[1] if len(param) > 1 else 0
    y = torch.empty((V, F), device=DEV)

    def check(v):
        agg.run(dev(x), y, 512, 1)
        ps, ix, tg, vs = orc.locality_schedule(ptr, idx, param[0], V, ng, v)
        assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, vs, x, V))
        assert np.array_equal(agg.get_schedule("scheduled", with_val=True)[3], vs)

    check(v1)
    dv.copy_(dev(v2))          # rewritten in place
    check(v2)
    dv3 = dev(v3)
    agg.updateval(dv3)         # re-aliased
    check(v3)


@pytest.mark.parametrize("kind,param", [("locality", [3]), ("locality_neighbor_grouping", [4, 8])])
def test_user_locality_schedule_that_drops_edges_is_not_replaced_by_the_balanced_order(kind, param):
    """The reference's locality schedulers keep only the edges with idx < total_num_v (graph_schedule.h:23-44): a schedule cut with
    total_num_v below the column count DROPS edges.  `scheduled = 1` must then run the user's groups even with the default
    "fast_scheduled" = 1 -- the balanced order covers every edge and would differ by whole edges, not by association (round-3
    advisor finding).  GCN and GAT; a schedule that keeps every edge still takes the balanced order."""
    V, E, F = 300, 9000, 64
    ptr, idx = gnc.graph.uniform_random_csr(V, E, seed=12)
    x, val = rand((V, F), 1), rand(E, 2)
    cut = 200                                         # columns 200 .. 299 are outside every range
    ng = param[1] if len(param) > 1 else 0
    ps, ix, tg, vs = orc.locality_schedule(ptr, idx, param[0], cut, ng, val)
    assert len(ix) < E
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)      # default options: fast_scheduled = 1
    agg.schedule(gnc.Schedule[kind], param, total_num_v=cut)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 512, 1)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, vs, x, V))
    full = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    full.schedule(gnc.Schedule[kind], param)          # total_num_v = num_v: every edge kept -> the balanced order
    yb = torch.empty((V, F), device=DEV)
    full.run(dev(x), y, 512, 1)
    full.run(dev(x), yb, 512, "balanced")
    assert torch.equal(y, yb)
    att = rand((V, 2), 3) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.schedule(gnc.Schedule[kind], param, total_num_v=cut)
    gat.run(dev(x), dev(att), y, 128, 1)
    ref, _, _ = orc.gat_grouped(ps, tg, ix, att, x, V, 1, seg=0)
    sub_ptr = np.concatenate([[0], np.cumsum(np.bincount(np.repeat(np.arange(V), np.diff(ptr))[idx < cut], minlength=V))]).astype(np.int32)
    assert_within(y.cpu().numpy(), ref, gat_scale(sub_ptr, idx[idx < cut], att, x, 1) + np.abs(ref), "gat on a schedule that drops edges")


def test_run_clock_checks_the_timer_capacity():
    """The size query without feature pointers assumes 16-byte lanes; a 4-byte-aligned x picks scalar lanes, more column
    tiles and a larger grid -- the timed call must refuse a buffer sized from the wrong answer (round-1 advisor finding)."""
    V, E, F = 2000, 30000, 128
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=21)
    val = torch.ones(E, device=DEV)
    agg = gnc.Aggregator_GCN(ptr_t.to(DEV), idx_t.to(DEV), val, F, F)
    L = gnc.lib()
    xb = torch.randn(V * F + 1, device=DEV)
    x_odd = xb[1:].view(V, F)
    y = torch.empty((V, F), device=DEV)
    nb_aligned, nb_odd = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(L.gnnagg_gcn_run_clock(agg._h, None, None, F, _lib.MODE_ROWS, None, ctypes.byref(nb_aligned), None))
    _lib.check(L.gnnagg_gcn_run_clock(agg._h, x_odd.data_ptr(), y.data_ptr(), F, _lib.MODE_ROWS, None, ctypes.byref(nb_odd), None))
    assert nb_odd.value > nb_aligned.value
    timer = torch.zeros((nb_odd.value, 3), dtype=torch.int64, device=DEV)
    cap = ctypes.c_int(nb_aligned.value)
    rc = L.gnnagg_gcn_run_clock(agg._h, x_odd.data_ptr(), y.data_ptr(), F, _lib.MODE_ROWS, timer.data_ptr(), ctypes.byref(cap), None)
    assert rc == _lib.ERR_ARG and b"timer buffer too small" in L.gnnagg_last_error()
    assert int(timer.abs().sum()) == 0                                # nothing was written
    cap = ctypes.c_int(nb_odd.value)
    _lib.check(L.gnnagg_gcn_run_clock(agg._h, x_odd.data_ptr(), y.data_ptr(), F, _lib.MODE_ROWS, timer.data_ptr(), ctypes.byref(cap), None))
    torch.cuda.synchronize()
    assert cap.value == nb_odd.value and int((timer[:, 1] != 0).sum()) > 0.9 * nb_odd.value


@pytest.mark.parametrize("case", ["chunked", "blocked", "scheduled"])
def test_gather_probe_runs_the_same_work_and_writes_nothing(case):
    V, E, F = 4000, 300000, 128
    ptr, idx = hub_graph(V, E, seed=13)
    x = rand((V, F), 1)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), torch.ones(E, device=DEV), F, F)
    mode = "balanced"
    if case == "blocked":
        agg.set_option("partitions", 4)
    elif case == "chunked":
        agg.set_option("partitions", 0)
    else:
        agg.set_option("fast_scheduled", 0)  # the user's groups in the restated order (the default runs the balanced order)
        agg.schedule(gnc.Schedule.neighbor_grouping, [256])   # few rows split: the plan kernel runs this schedule
        mode = "scheduled"
        assert agg.mode_params("scheduled") == (256, 16)
    dx = dev(x)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dx, y, 128, mode)
    ref = y.clone()
    agg.probe_gather(dx, mode)
    torch.cuda.synchronize()
    assert torch.equal(dx.cpu(), torch.from_numpy(x)) and torch.equal(y, ref)
    with pytest.raises(gnc.GnnAggError):
        agg.probe_gather(dx, "rows")


def test_gat_probe_runs_on_the_blocked_order_only_and_writes_nothing():
    V, E, F, H = 800, 200000, 128, 4
    ptr, idx = hub_graph(V, E, seed=9)
    x, att = dev(rand((V, F), 1)), dev(rand((V, H, 2), 2) * 0.4)
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("slice_kb", 16)
    assert gat.balanced_partitions() > 1
    x0, att0 = x.clone(), att.clone()
    y = torch.full((V, F), 7.0, device=DEV)
    gat.run(x, att, y, 128, "balanced", heads=H)
    y0 = y.clone()
    gat.probe_gather(x, att, "balanced", heads=H)
    torch.cuda.synchronize()
    assert torch.equal(x, x0) and torch.equal(att, att0)
    gat.run(x, att, y, 128, "balanced", heads=H)   # scratch untouched by the probe: same bits again
    assert torch.equal(y, y0)
    with pytest.raises(Exception):
        gat.probe_gather(x, att, "rows", heads=H)
    chunked = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    chunked.set_option("partitions", 0)   # the chunked plan has no probe instantiation
    with pytest.raises(Exception):
        chunked.probe_gather(x, att, "balanced", heads=H)


def test_set_option_rejects_unknown_names_and_values():
    ptr, idx = gnc.graph.uniform_random_csr(50, 400, seed=1)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, 32, 32)
    for name, value in (("no_such_option", 1), ("tile_width", 48), ("slice_kb", 0), ("partitions", -2)):
        with pytest.raises(gnc.GnnAggError):
            agg.set_option(name, value)


def test_hub_fold_stress_two_handles_two_streams():
    """The in-kernel hub fold under load: segment workgroups of dozens of hubs publish their sums across XCDs (sc1 stores,
    drained, one agent-scope atomic per workgroup) while a second handle does the same on another stream; hundreds of
    launches alternating three inputs, every checked result bit-equal to the first launch's and to the oracle."""
    V, E, F = 20000, 1500000, 128
    ptr, idx = gnc.graph.powerlaw_csr(V, E, seed=3, alpha=1.1, device=DEV)
    val = torch.randn(E, device=DEV)
    aggs = [gnc.Aggregator_GCN(ptr, idx, val, F, F) for _ in range(2)]
    for a in aggs:
        a.set_option("partitions", 0)
        a.schedule_balanced(16)      # 256-edge segments: dozens of hubs with tens to hundreds of segments
    deg = (ptr[1:] - ptr[:-1])
    assert int((deg > 4096).sum()) >= 10
    xs = [torch.randn((V, F), device=DEV) for _ in range(3)]
    refs = []
    y = torch.empty((V, F), device=DEV)
    for x in xs:
        aggs[0].run(x, y, 128, "balanced")
        refs.append(y.clone())
    chunk, seg = aggs[0].balanced_params()
    ps, tg = orc.neighbor_grouping(ptr.cpu().numpy(), chunk)
    assert np.array_equal(refs[0].cpu().numpy(),
                          orc.gcn_grouped(ps, tg, idx.cpu().numpy(), val.cpu().numpy(), xs[0].cpu().numpy(), V, seg=seg))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ys = [torch.empty((V, F), device=DEV) for _ in range(2)]
    torch.cuda.synchronize()
    bad, N = 0, 600
    for it in range(N):
        k = it % 3
        for a, st, yy in zip(aggs, streams, ys):
            with torch.cuda.stream(st):
                a.run(xs[k], yy, 128, "balanced")
        if it % 25 == 0 or it > N - 4:
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(yy, refs[k]) else 1 for yy in ys)
    torch.cuda.synchronize()
    assert bad == 0


def test_blocked_order_demotes_to_the_chunked_plan_when_its_scratch_does_not_fit():
    """The blocked order needs one partial row per group and a tiled image of X; a handle whose scratch would not fit (here:
    a 1 MB cap) moves to the chunked plan for good, reports that order from then on, and stays correct (round-1 advisor
    finding: the description of the order has to follow the demotion)."""
    V, E, F = 900, 260000, 128
    ptr, idx = hub_graph(V, E, seed=5)
    x, val = rand((V, F), 1), rand(E, 2)
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    agg.set_option("slice_kb", 16)
    assert agg.balanced_partitions() > 1
    agg.set_option("scratch_limit_mb", 1)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    assert agg.balanced_partitions() == 0
    chunk, seg = agg.balanced_params()
    assert seg == 16
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, idx, val, x, V, seg=seg))
    got = agg.get_schedule("balanced")
    assert np.array_equal(got[0], ps) and np.array_equal(got[2], tg)
    agg.run(dev(x), y, 128, "balanced", reduce="mean")       # and it stays there
    assert agg.balanced_partitions() == 0
    # GAT handles demote the same way
    att = rand((V, 2), 3) * 0.4
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("slice_kb", 16)
    gat.set_option("scratch_limit_mb", 1)
    gat.run(dev(x), dev(att), y, 128, "balanced")
    assert gat.balanced_partitions() == 0
    ch, sg = gat.balanced_params()
    ref, _, _ = orc.gat_grouped(*orc.neighbor_grouping(ptr, ch), idx, att, x, V, 1, seg=sg)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, 1) + np.abs(ref), "demoted gat")


def test_forced_partition_count_is_clamped_to_the_column_count():
    """"partitions" / GNNAGG_PARTITIONS above the number of columns used to inflate total_cols, and the tiling kernels
    then read that many rows of the caller's X (and att) -- past their ends.  The count is clamped to the columns that occur."""
    V, F = 300, 64
    rng = np.random.default_rng(5)
    deg = rng.integers(1, 40, V)
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    idx = np.sort(rng.integers(0, 3, int(ptr[-1]))).astype(np.int32)          # only columns 0..2 occur
    for r in range(V):
        idx[ptr[r]:ptr[r + 1]].sort()
    x = rand((3, F), 1)                                                         # X has exactly as many rows as there are columns
    agg = gnc.Aggregator_GCN(dev(ptr), dev(idx), None, F, F)
    agg.set_option("partitions", 8)
    y = torch.full((V, F), 7.0, device=DEV)
    agg.run(dev(x), y, 128, "balanced")
    parts, cols = agg.balanced_partitions(), agg.balanced_partition_columns()
    assert 1 <= parts <= 3 and cols == 3
    ps, ix, tg, _ = orc.locality_schedule(ptr, idx, parts, cols, ng=agg.balanced_params()[0])
    assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, None, x, V, seg=0))
    att = rand((V, 2), 3) * 0.3                                                 # GAT: att has V rows (> columns), X 3 rows
    gat = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gat.set_option("partitions", 8)
    xg = np.zeros((V, F), np.float32)
    xg[:3] = x
    gat.run(dev(xg), dev(att), y, 128, "balanced")
    ref = orc.gat_fused(ptr, idx, att, xg)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, xg, 1) + np.abs(ref), "clamped partitions, gat")


def _split_edges(ptr, idx, val, seed, frac=0.4):
    """two CSRs over the same rows: a random `frac` of every row's edges and the rest, in-row order kept"""
    rng = np.random.default_rng(seed)
    first = rng.random(len(idx)) < frac
    rows = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))

    def sub(mask):
        p = np.zeros(len(ptr), np.int32)
        p[1:] = np.cumsum(np.bincount(rows[mask], minlength=len(ptr) - 1))
        return p, idx[mask].astype(np.int32), (None if val is None else val[mask])
    return sub(first), sub(~first)


@pytest.mark.parametrize("F", [128, 100, 30])
@pytest.mark.parametrize("chunk", [0, 8])
def test_two_pass_mean_and_max_with_row_aux_on_hub_rows(F, chunk):
    """gnnagg_set_row_aux (the row-partitioned step's two passes) on every finishing site of the plan kernel: short rows,
    single-segment long rows, hubs folded by the last segment workgroup to arrive, and -- with the in-kernel fold switched off -- the
    ordered combine.  Mean: both passes divide by the row's TOTAL degree and add; max: the second pass joins only where the first
    folded an edge.  Restated exactly with the oracle's grouped fold of each half."""
    V, E = 3000, 120000
    ptr, idx = hub_graph(V, E, seed=21, alpha=1.0)      # rows with thousands of edges: segments and hubs in both halves
    val, x = rand(E, 2), rand((V, F), 1)
    (pa, ia, va), (pb, ib, vb) = _split_edges(ptr, idx, val, seed=5)
    deg_tot, deg_a = dev(np.diff(ptr).astype(np.int32)), dev(np.diff(pa).astype(np.int32))
    for inkernel in ((1, 0) if _gl.has_extras() else (1,)):
        a = gnc.Aggregator_GCN(dev(pa), dev(ia), dev(va), F, F)
        b = gnc.Aggregator_GCN(dev(pb), dev(ib), dev(vb), F, F)
        for h in (a, b):
            if _gl.has_extras():
                h.set_option("inkernel_combine", inkernel)
            h.set_option("partitions", 0)
            h.schedule_balanced(chunk)
        ca, sa = a.balanced_params()
        cb, sb = b.balanced_params()
        ya = orc.gcn_grouped(*orc.neighbor_grouping(pa, ca), ia, va, x, V, seg=sa)
        yb = orc.gcn_grouped(*orc.neighbor_grouping(pb, cb), ib, vb, x, V, seg=sb)
        has_b = (np.diff(pb) > 0)[:, None]
        d = np.maximum(np.diff(ptr), 1)[:, None].astype(np.float32)
        y = torch.full((V, F), 7.0, device=DEV)
        a.set_row_aux(deg_tot)
        b.set_row_aux(deg_tot)
        a.run(dev(x), y, 512, "balanced", reduce="mean")
        b.run(dev(x), y, 512, "balanced", reduce="mean", accumulate=True)
        has_a = (np.diff(pa) > 0)[:, None]
        want = np.where(has_a, ya / d, 0.0).astype(np.float32)
        want = np.where(has_b, want + (yb / d).astype(np.float32), want)
        assert np.array_equal(y.cpu().numpy(), want), ("mean", inkernel)
        assert_within(y.cpu().numpy(), orc.gcn_seq(ptr, idx, val, x) / d, orc.gcn_abs_scale(ptr, idx, val, x) / d, "two-pass mean")
        a.set_row_aux(None)
        b.set_row_aux(deg_a)
        a.run(dev(x), y, 512, "balanced", reduce="max")
        b.run(dev(x), y, 512, "balanced", reduce="max", accumulate=True)
        assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, val, x)), ("max", inkernel)
        b.set_row_aux(None)      # without the array a mean / max accumulate is refused, a sum is what it always was
        with pytest.raises(Exception):
            b.run(dev(x), y, 512, "balanced", reduce="max", accumulate=True)
        a.run(dev(x), y, 512, "balanced")
        b.run(dev(x), y, 512, "balanced", accumulate=True)
        assert np.array_equal(y.cpu().numpy(), np.where(has_b, ya + yb, ya))


@pytest.mark.parametrize("F,H", [(128, 1), (256, 8), (64, 2)])
def test_two_pass_gat_on_hub_rows(F, H):
    """gnnagg_gat_run_part: numerators and denominators of one half of every row's edges, then the other half added and the rows
    divided -- on short rows, segments and hubs -- against the fused single pass and the oracle."""
    V, E = 3000, 120000
    ptr, idx = hub_graph(V, E, seed=22, alpha=1.0)
    x, att = rand((V, F), 1), rand((V, H, 2), 3) * 0.4
    (pa, ia, _), (pb, ib, _) = _split_edges(ptr, idx, None, seed=6)
    a, b = gnc.Aggregator_GAT(dev(pa), dev(ia), F, F), gnc.Aggregator_GAT(dev(pb), dev(ib), F, F)
    y, den = torch.full((V, F), 7.0, device=DEV), torch.full((V, H), 7.0, device=DEV)
    a.run_part(dev(x), dev(att), y, den, 1, heads=H)
    _, _, (na, da) = orc.gat_grouped(*orc.neighbor_grouping(pa, a.balanced_params()[0]), ia, att, x, V, H, seg=a.balanced_params()[1], parts=True)
    assert np.allclose(den.cpu().numpy(), da, rtol=1e-5, atol=1e-6)          # part 1 leaves raw numerators / denominators
    num_scale = (gat_scale(pa, ia, att, x, H).astype(np.float64) * np.repeat(da, F // H, axis=1)).astype(np.float32)   # sum_e w_e |x_e|
    assert_within(y.cpu().numpy(), na.astype(np.float32), 2 * num_scale + 1e-6, "two-pass gat, numerators")
    b.run_part(dev(x), dev(att), y, den, 2, heads=H)
    ref = orc.gat_fused(ptr, idx, att, x, H)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "two-pass gat")
    assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0)
    # part 3 (a pass in between: both added, nothing divided -- the staged halo exchange): a (1), b (3), then a third handle
    # without edges (2) divides; the same sums in the same order as a (1), b (2): the same bits
    y2, den2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, H), 7.0, device=DEV)
    none = gnc.Aggregator_GAT(dev(np.zeros(V + 1, np.int32)), dev(np.zeros(0, np.int32)), F, F)
    a.run_part(dev(x), dev(att), y2, den2, 1, heads=H)
    b.run_part(dev(x), dev(att), y2, den2, 3, heads=H)
    none.run_part(dev(x), dev(att), y2, den2, 2, heads=H)
    assert torch.equal(y, y2)
    with pytest.raises(Exception):
        a.run_part(dev(x), dev(att), y, den, 4, heads=H)


@pytest.mark.parametrize("F", [602, 100, 64, 256, 33])
@pytest.mark.parametrize("slice_kb", [16, 4])
def test_rows_mode_on_the_blocked_order_is_the_canonical_chain(F, slice_kb):
    """GNNAGG_MODE_ROWS (`scheduled = 0`) of a GCN handle on a graph the blocked order applies to (average degree >= 96, neighbors
    ascending in every row): the reference's locality_schedule groups (one per (row, source range)) walked one range per launch, every
    chain carried from range to range through a tiled image of Y -- range after range is the CSR order, so the result is the
    reference's one sequential chain per (row, column) (aggr_gcn.h:13-35): bit-equal to the oracle's gcn_seq / gcn_mean and to the row
    kernels, explicit and implicit weights, ReLU, updateval, hub rows, rows without edges.  Unsorted rows and max keep the row kernels."""
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=5)
    assert all(np.all(np.diff(idx[ptr[r]:ptr[r + 1]]) >= 0) for r in range(V))   # the generator sorts the rows
    x, val = rand((V, F), 1), rand(E, 2)
    for v in (val, None):
        a = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if v is None else dev(v), F, F)
        a.set_option("slice_kb", slice_kb)
        k = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if v is None else dev(v), F, F)
        k.set_option("rows_blocked", 0)
        assert a.rows_blocked_ranges() > 1 and k.rows_blocked_ranges() == 0
        y, y2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, F), 7.0, device=DEV)
        a.run(dev(x), y, 128, 0)
        ref = orc.gcn_seq(ptr, idx, v, x)
        assert np.array_equal(y.cpu().numpy(), ref)
        assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0)
        a.run(dev(x), y, 128, 0, reduce="mean")
        assert np.array_equal(y.cpu().numpy(), orc.gcn_mean(ptr, idx, v, x))
        for kw in ({}, {"reduce": "mean"}, {"relu": True}, {"reduce": "max"}):
            a.run(dev(x), y, 128, 0, **kw)
            k.run(dev(x), y2, 128, 0, **kw)
            assert torch.equal(y, y2), kw
        one = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if v is None else dev(v), F, F)   # hub rows behind the launches, one stream
        one.set_option("slice_kb", slice_kb)
        one.set_option("aux_stream", 0)
        one.run(dev(x), y2, 128, 0, reduce="mean")
        a.run(dev(x), y, 128, 0, reduce="mean")
        assert torch.equal(y, y2)
        if v is not None:
            v2 = rand(E, 9)
            a.updateval(dev(v2))
            a.run(dev(x), y, 128, 0)
            assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx, v2, x))
    # rows that do not list their neighbors in ascending order: range after range is not the CSR order -- the row kernels run
    rng = np.random.default_rng(3)
    idx_u = idx.copy()
    for r in range(0, V, 7):
        rng.shuffle(idx_u[ptr[r]:ptr[r + 1]])
    u = gnc.Aggregator_GCN(dev(ptr), dev(idx_u), dev(val), F, F)
    u.set_option("slice_kb", slice_kb)
    assert u.rows_blocked_ranges() == 0
    y = torch.full((V, F), 7.0, device=DEV)
    u.run(dev(x), y, 128, 0)
    assert np.array_equal(y.cpu().numpy(), orc.gcn_seq(ptr, idx_u, val, x))


@pytest.mark.parametrize("F,H", [(256, 8), (64, 1), (128, 2), (128, 1), (96, 3), (192, 2), (512, 16)])
@pytest.mark.parametrize("slice_kb", [16, 4])
def test_gat_rows_mode_on_the_blocked_order_is_the_canonical_chain(F, H, slice_kb):
    """GNNAGG_MODE_ROWS of a GAT handle (`scheduled = 0` = aggr_gat, aggr_gat.h:116-164) on a graph the blocked order applies to: the
    numerator chain of every (row, column) and the denominator chain of every (row, head) carried from source range to source range
    through tiled images (k_gat_span<..., CHAIN>), one division at the end.  Range after range is the CSR order,
    so it is the same chain the row kernels run: bit-equal to them ("rows_blocked" = 0), within the 1e-5 bound of the oracle's fused
    result; heads narrower than, equal to and wider than the 64-float tile, a ragged last tile, hub rows on the workgroup-per-row
    kernel beside the launches or behind them, rows without edges 0; callers that ask for newval keep the row kernels."""
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=5)
    x, att = rand((V, F), 1), rand((V, H, 2), 2) * 0.4
    a = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    a.set_option("slice_kb", slice_kb)
    k = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    k.set_option("rows_blocked", 0)
    y, y2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, F), 7.0, device=DEV)
    for _ in range(2):   # (the second call reuses the images: they are zeroed by every run)
        a.run(dev(x), dev(att), y, 128, 0, heads=H)
    k.run(dev(x), dev(att), y2, 128, 0, heads=H)
    assert a.rows_blocked_ranges() > 1 and k.rows_blocked_ranges() == 0
    ref = orc.gat_fused(ptr, idx, att, x, H)
    assert_within(y.cpu().numpy(), ref, gat_scale(ptr, idx, att, x, H) + np.abs(ref), "gat chains on the blocked order")
    assert np.all(y.cpu().numpy()[np.diff(ptr) == 0] == 0)
    assert torch.equal(y, y2), "the chained form and the row kernels run the same chains"
    for aux in (0, 1):   # rows with a sub-row above the threshold leave the chains: whole, on the workgroup-per-row kernel (head widths % 32 == 0)
        one = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
        one.set_option("slice_kb", slice_kb)
        one.set_option("rows_hub_edges", 150)
        one.set_option("aux_stream", aux)
        y2.fill_(7.0)
        one.run(dev(x), dev(att), y2, 128, 0, heads=H)
        assert one.rows_blocked_ranges() > 1 and torch.equal(y, y2), aux
    nv, nv2 = torch.full((E, H), 7.0, device=DEV), torch.full((E, H), 7.0, device=DEV)
    a.run(dev(x), dev(att), y2, 128, 0, heads=H, newval=nv)
    k.run(dev(x), dev(att), y, 128, 0, heads=H, newval=nv2)
    assert torch.equal(nv, nv2) and torch.equal(y, y2)


def test_rows_mode_on_the_blocked_order_with_the_dense_combine_behind_it():
    """run_with_nn in the canonical order on a graph whose chains run on the blocked order: vout is the sequential chain, transformed
    the oracle's ascending-k GEMM of it, both bit for bit; and the whole run replays from a captured HIP graph (memset, one launch per
    range, the hub rows forked to the auxiliary stream and joined)."""
    V, E, F, OUT = 900, 260000, 128, 32
    ptr, idx = hub_graph(V, E, seed=5)
    x, val, W = rand((V, F), 1), rand(E, 2), rand((F, OUT), 3) * 0.3
    a = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, OUT)
    a.set_option("slice_kb", 16)
    assert a.rows_blocked_ranges() > 1
    y, tr = torch.full((V, F), 7.0, device=DEV), torch.full((V, OUT), 7.0, device=DEV)
    a.run_with_nn(dev(x), y, dev(W), tr, 128, 0)
    ref = orc.gcn_seq(ptr, idx, val, x)
    assert np.array_equal(y.cpu().numpy(), ref)
    assert np.array_equal(tr.cpu().numpy(), orc.matmul_nn(ref, W))
    dx = dev(x)
    a.run(dx, y, 128, 0)           # warm: plan, scratch, streams and events exist
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    y.fill_(7.0)
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            a.run(dx, y, 128, 0, reduce="mean")
    for _ in range(2):
        y.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(y.cpu().numpy(), orc.gcn_mean(ptr, idx, val, x))
