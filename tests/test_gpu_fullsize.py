"""GPU, BASELINE.json's full sizes: the other configs (reddit-shaped SAGE mean F=602, reddit-shaped GAT
8 heads x 32, products-shaped GCN F=100) checked through size-independent properties and against the
oracle on a random sample of rows (the oracle cannot finish 115 M edges x 602 columns in seconds, but a
row sample is exact: per-row results depend only on that row's edges)."""
import numpy as np
import pytest
import torch

import gnn_computing_amd as gnc
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sample_rows(ptr, idx, rows):
    """Sub-CSR of the chosen rows (numpy), column ids unchanged."""
    ptr_h = ptr.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, np.int32)
    sub_ptr[1:] = np.cumsum(ptr_h[rows + 1] - ptr_h[rows])
    parts = [idx[int(ptr_h[r]):int(ptr_h[r + 1])] for r in rows]
    sub_idx = torch.cat(parts).cpu().numpy() if parts else np.empty(0, np.int32)
    return sub_ptr, sub_idx, np.concatenate([np.arange(ptr_h[r], ptr_h[r + 1]) for r in rows]).astype(np.int64)


def pick_rows(ptr, k, seed):
    """k random rows + the heaviest row + an empty row."""
    deg = (ptr[1:] - ptr[:-1]).cpu().numpy()
    rng = np.random.default_rng(seed)
    rows = set(rng.integers(0, len(deg), k).tolist())
    rows.add(int(deg.argmax()))
    empties = np.nonzero(deg == 0)[0]
    if len(empties):
        rows.add(int(empties[0]))
    return np.array(sorted(rows), np.int64)


@pytest.fixture(scope="module")
def reddit():
    ptr, idx = gnc.graph.dataset("reddit", device=DEV)
    assert (ptr.numel() - 1, idx.numel()) == gnc.graph.SHAPES["reddit"]
    return ptr, idx


def test_reddit_sage_mean_f602(reddit):
    ptr, idx = reddit
    V, E, F = ptr.numel() - 1, idx.numel(), 602
    agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    x = torch.randn((V, F), device=DEV)
    y = torch.empty((V, F), device=DEV)
    agg.run(x, y, 512, "balanced", reduce="mean")
    rows = pick_rows(ptr, 40, 1)
    sp, si, _ = sample_rows(ptr, idx, rows)
    xh = x.cpu().numpy()
    chunk, seg = agg.balanced_params()
    parts = agg.balanced_partitions()
    assert parts >= 8 and seg == 0   # avg degree 492: the library picks the source-partitioned (2-D blocked) order here
    ps, ix, tg, _ = orc.locality_schedule(sp, si, parts, agg.balanced_partition_columns(), ng=chunk)   # the same order restated on the sampled rows
    ref_sum = orc.gcn_grouped(ps, tg, ix, None, xh, len(rows), seg=0)
    deg = np.maximum(np.diff(sp), 1)[:, None].astype(np.float32)
    got = y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    assert np.array_equal(got, ref_sum / deg)  # same partial order, same final IEEE division
    # against the canonical CSR-order mean within the fp32 bound
    ref = orc.gcn_mean(sp, si, None, xh)
    scale = orc.gcn_abs_scale(sp, si, None, xh) / deg
    assert np.all(np.abs(got - ref) <= 1e-5 * scale + 1e-30)
    # size-independent: the mean of a constant is that constant (exactly), empty rows are 0
    ones = torch.full((V, F), 3.0, device=DEV)
    agg.run(ones, y, 512, "balanced", reduce="mean")
    degs = (ptr[1:] - ptr[:-1])
    assert bool(torch.all(y[degs > 0] == 3.0)) and bool(torch.all(y[degs == 0] == 0.0))
    # sum of ones = degree, exact in fp32 (max degree < 2^24)
    agg.run(ones.fill_(1.0), y, 512, "balanced", reduce="sum")
    assert bool(torch.all(y[:, 0] == degs.to(torch.float32))) and bool(torch.all(y[:, 601] == degs.to(torch.float32)))


def test_reddit_sage_mean_f602_canonical_rows_mode(reddit):
    """BASELINE's SAGE case in the canonical order (`scheduled = 0`): on this graph (average degree 492, sorted rows) the chains run
    on the 2-D blocked order, one source range per launch.  Sampled rows bit-equal to the oracle's sequential chain (the reference's
    aggr_gcn order, aggr_gcn.h:13-35) -- hub rows included --, the whole output bit-equal to the row kernels, and the
    size-independent identities of the balanced test."""
    ptr, idx = reddit
    V, E, F = ptr.numel() - 1, idx.numel(), 602
    agg = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    assert agg.rows_blocked_ranges() >= 8
    x = torch.randn((V, F), device=DEV)
    y = torch.empty((V, F), device=DEV)
    agg.run(x, y, 512, 0, reduce="mean")
    degs = (ptr[1:] - ptr[:-1])
    rows = np.unique(np.concatenate([pick_rows(ptr, 40, 1), torch.argsort(degs, descending=True)[:3].cpu().numpy()]))   # + the three largest hubs
    sp, si, _ = sample_rows(ptr, idx, rows)
    got = y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    assert np.array_equal(got, orc.gcn_mean(sp, si, None, x.cpu().numpy()))
    kern = gnc.Aggregator_GCN(ptr, idx, None, F, F)
    kern.set_option("rows_blocked", 0)
    y2 = torch.empty((V, F), device=DEV)
    kern.run(x, y2, 512, 0, reduce="mean")
    assert torch.equal(y, y2)
    ones = torch.full((V, F), 3.0, device=DEV)
    agg.run(ones, y, 512, 0, reduce="mean")
    assert bool(torch.all(y[degs > 0] == 3.0)) and bool(torch.all(y[degs == 0] == 0.0))
    agg.run(ones.fill_(1.0), y, 512, 0, reduce="sum")
    assert bool(torch.all(y[:, 0] == degs.to(torch.float32))) and bool(torch.all(y[:, 601] == degs.to(torch.float32)))


def test_reddit_gat_8x32(reddit):
    ptr, idx = reddit
    V, E, H, D = ptr.numel() - 1, idx.numel(), 8, 32
    F = H * D
    gat = gnc.Aggregator_GAT(ptr, idx, F, F)
    x = torch.randn((V, F), device=DEV)
    att = torch.randn((V, H, 2), device=DEV) * 0.5
    y = torch.empty((V, F), device=DEV)
    gat.run(x, att, y, 128, "balanced", heads=H)
    rows = pick_rows(ptr, 40, 2)
    sp, si, _ = sample_rows(ptr, idx, rows)
    xh, atth = x.cpu().numpy(), att.cpu().numpy()
    # The library's order restated on the sampled rows (as for SAGE above): the groups of the source-partitioned schedule,
    # folded flat in ascending group order.  The oracle reads the centre term of compact row k at att[k,:,0] and the source
    # terms at att[id,:,1] -- different slots, so one array can carry both.
    chunk, seg = gat.balanced_params()
    parts = gat.balanced_partitions()
    assert parts >= 8 and seg == 0
    att_mix = atth.copy()
    att_mix[:len(rows), :, 0] = atth[rows, :, 0]
    ps, ix, tg, _ = orc.locality_schedule(sp, si, parts, gat.balanced_partition_columns(), ng=chunk)
    ref, _, _ = orc.gat_grouped(ps, tg, ix, att_mix, xh, len(rows), H, seg=0)
    got = y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    # north_star's 1e-5, condition-aware: same association as the kernel, device expf vs libm is what is left
    w = orc.gat_att(sp, si, att_mix, H)                       # normalised weights of the sampled rows' edges [E', H]
    scale = np.zeros((len(rows), F))
    for k in range(len(rows)):
        e0, e1 = int(sp[k]), int(sp[k + 1])
        if e1 > e0:
            scale[k] = np.einsum("eh,ehd->hd", w[e0:e1].astype(np.float64),
                                 np.abs(xh[si[e0:e1]]).reshape(e1 - e0, H, D)).reshape(F)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    bound = 1e-5 * (scale + np.abs(ref)) + 1e-30
    assert np.all(err <= bound), "GAT 8x32: worst ratio %.3g" % float((err / bound).max())
    # and against the canonical CSR-order fused result (different association): same bound
    ref_csr = orc.gat_fused(sp, si, att_mix, xh, H)
    err = np.abs(got.astype(np.float64) - ref_csr.astype(np.float64))
    assert np.all(err <= bound), "GAT 8x32 vs CSR order: worst ratio %.3g" % float((err / bound).max())
    assert not np.isnan(got).any()
    # size-independent: softmax weights sum to 1 -> aggregating a constant gives that constant (1e-5)
    const = torch.full((V, F), 2.0, device=DEV)
    gat.run(const, att, y, 128, "balanced", heads=H)
    degs = (ptr[1:] - ptr[:-1])
    nz = y[degs > 0]
    assert float((nz - 2.0).abs().max()) <= 2.0 * 1e-5 and bool(torch.all(y[degs == 0] == 0.0))
    # adapter path == fused path (Figure10/main_a.cu:98-100) on one head
    gat1 = gnc.Aggregator_GAT(ptr, idx, D, D)
    x1, att1 = x[:, :D].contiguous(), att[:, 0, :].contiguous()
    y1, y2 = torch.empty((V, D), device=DEV), torch.empty((V, D), device=DEV)
    gat1.run(x1, att1, y1, 128, "balanced")
    w = torch.empty(E, device=DEV)
    gat1.run_att(att1, w, 128)
    gcn = gnc.Aggregator_GCN(ptr, idx, w, D, D)
    gcn.run(x1, y2, 128, "balanced")
    # two routes to the same value, each within 1e-5 * sum_e w_e |x_e| of it
    scale = torch.empty((V, D), device=DEV)
    gcn.run(x1.abs(), scale, 128, "balanced")
    assert bool(torch.all((y1 - y2).abs() <= 2e-5 * scale + 1e-30))


def test_reddit_gat_8x32_canonical_rows_mode(reddit):
    """Config G in GNNAGG_MODE_ROWS (`scheduled = 0` = aggr_gat, aggr_gat.h:116-164) at full size: the chains run on the 2-D blocked order
    (k_gat_span<..., CHAIN>) -- sampled rows against the oracle's CSR-order fused result within north_star's 1e-5
    (condition-aware), every row bit-equal to the row kernels ("rows_blocked" = 0), constants reproduced, rows without edges 0."""
    ptr, idx = reddit
    V, H, D = ptr.numel() - 1, 8, 32
    F = H * D
    gat = gnc.Aggregator_GAT(ptr, idx, F, F)
    x = torch.randn((V, F), device=DEV)
    att = torch.randn((V, H, 2), device=DEV) * 0.5
    y = torch.empty((V, F), device=DEV)
    gat.run(x, att, y, 128, 0, heads=H)
    assert gat.rows_blocked_ranges() >= 4
    rows = pick_rows(ptr, 40, 3)
    sp, si, _ = sample_rows(ptr, idx, rows)
    xh, atth = x.cpu().numpy(), att.cpu().numpy()
    att_mix = atth.copy()
    att_mix[:len(rows), :, 0] = atth[rows, :, 0]
    ref = orc.gat_fused(sp, si, att_mix, xh, H)
    got = y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    w = orc.gat_att(sp, si, att_mix, H)
    scale = np.zeros((len(rows), F))
    for k in range(len(rows)):
        e0, e1 = int(sp[k]), int(sp[k + 1])
        if e1 > e0:
            scale[k] = np.einsum("eh,ehd->hd", w[e0:e1].astype(np.float64), np.abs(xh[si[e0:e1]]).reshape(e1 - e0, H, D)).reshape(F)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    bound = 1e-5 * (scale + np.abs(ref)) + 1e-30
    assert np.all(err <= bound), "GAT 8x32 rows mode: worst ratio %.3g" % float((err / bound).max())
    kern = gnc.Aggregator_GAT(ptr, idx, F, F)
    kern.set_option("rows_blocked", 0)
    y2 = torch.empty((V, F), device=DEV)
    kern.run(x, att, y2, 128, 0, heads=H)
    assert torch.equal(y, y2), "the chained form and the row kernels run the same chains"
    const = torch.full((V, F), 2.0, device=DEV)
    gat.run(const, att, y, 128, 0, heads=H)
    degs = (ptr[1:] - ptr[:-1])
    assert float((y[degs > 0] - 2.0).abs().max()) <= 2.0 * 1e-5 and bool(torch.all(y[degs == 0] == 0.0))


def test_products_gcn_f100():
    ptr, idx = gnc.graph.dataset("products", device=DEV)
    V, E, F = ptr.numel() - 1, idx.numel(), 100
    assert (V, E) == gnc.graph.SHAPES["products"]
    val = torch.rand(E, device=DEV) + 0.5
    agg = gnc.Aggregator_GCN(ptr, idx, val, F, F)
    x = torch.randn((V, F), device=DEV)
    y = torch.empty((V, F), device=DEV)
    agg.run(x, y, 512, "balanced")
    rows = pick_rows(ptr, 200, 3)
    sp, si, eids = sample_rows(ptr, idx, rows)
    xh = x.cpu().numpy()
    vh = val[torch.from_numpy(eids).to(DEV)].cpu().numpy()
    chunk, seg = agg.balanced_params()
    ps, tg = orc.neighbor_grouping(sp, chunk)
    got = y[torch.from_numpy(rows).to(DEV)].cpu().numpy()
    assert np.array_equal(got, orc.gcn_grouped(ps, tg, si, vh, xh, len(rows), seg=seg))
    # linearity (checksum of checksums): A(2x) == 2 A(x) exactly (power-of-two scaling commutes with rounding)
    y2 = torch.empty_like(y)
    agg.run(x * 2.0, y2, 512, "balanced")
    assert bool(torch.all(y2 == 2.0 * y))


def test_offsets_beyond_2_pow_31_elements():
    """Maximum sizes: feature matrices of more than 2^31 elements (4.3 M rows x 512 floats = 8.8 GB each), so every row offset
    above row 4 194 304 needs 64-bit arithmetic.  Two neighbors per row with unit weights: fma(x1, 1, fma(x0, 1, 0)) = x0 + x1
    exactly, whatever the order -- the check is a torch gather-add on the device, bit for bit, for the canonical rows mode, the
    balanced order and the mean."""
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs ~45 GB of device memory")
    V, F = 4_300_000, 512
    assert V * F > (1 << 31)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    idx2 = torch.randint(0, V, (V, 2), generator=g, device=DEV, dtype=torch.int32)
    idx2[:, 1] = (idx2[:, 1] // 2 + V // 2).clamp_(max=V - 1)     # every row reads at least one source beyond the 2^31-element mark
    idx2, _ = torch.sort(idx2, dim=1)
    ptr = torch.arange(0, 2 * V + 1, 2, dtype=torch.int32, device=DEV)
    x = torch.randn((V, F), generator=g, device=DEV)
    agg = gnc.Aggregator_GCN(ptr, idx2.reshape(-1).contiguous(), None, F, F)
    y = torch.full((V, F), 7.0, device=DEV)
    ref = torch.empty_like(x)
    step = 1 << 20
    for r0 in range(0, V, step):
        r1 = min(V, r0 + step)
        ref[r0:r1] = x[idx2[r0:r1, 0].long()] + x[idx2[r0:r1, 1].long()]
    for mode in (0, "balanced"):
        y.fill_(7.0)
        agg.run(x, y, 128, mode)
        assert torch.equal(y, ref), mode
    agg.run(x, y, 128, "balanced", reduce="mean")
    assert torch.equal(y, ref / 2)
