"""Tests of the two forms taken out of libgnnagg.so in round 4 (destination-stationary 2-D blocked order, agg_ds.hip; per-tile /
per-chunk combine overlap).  They ran green in GPUTEST_r03 against the library of round 3; kept with the code they tested
(scripts/attic/agg_ds.hip, scripts/attic/shed_r04.patch).  Not collected by pytest."""
@pytest.mark.parametrize("F", [602, 100, 64, 30, 256])
@pytest.mark.parametrize("slice_kb,hub_edges", [(16, 4096), (4, 4096), (16, 300), (4, 60)])
def test_destination_stationary_form_keeps_the_restated_order(F, slice_kb, hub_edges):
    """Option "dest_stationary" (agg_ds.hip; an experiment of round 3, default off): the groups of the 2-D blocked order with the
    accumulators of a unit's rows resident in LDS and the source ranges swept as phases -- no partial rows, no combine pass.  Same
    groups, same ascending fold: bit-equal to the streaming form and to the oracle's restatement, sum / mean / ReLU, explicit and
    implicit weights, hub sub-rows that continue across lane groups (staged continuation groups), hub rows routed to the streaming
    form beside the units, rows without edges."""
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=5)
    x, val = rand((V, F), 1), rand(E, 2)
    for v in (val, None):
        ds = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if v is None else dev(v), F, F)
        ds.set_option("slice_kb", slice_kb)
        ds.set_option("dest_stationary", 1)
        ds.set_option("ds_hub_edges", hub_edges)     # rows with a heavier sub-row stay on the streaming form (their own spans + combine)
        st = gnc.Aggregator_GCN(dev(ptr), dev(idx), None if v is None else dev(v), F, F)
        st.set_option("slice_kb", slice_kb)
        assert ds.balanced_partitions() == st.balanced_partitions() > 1
        ps, ix, tg, vs = blocked_reference(ds, ptr, idx, v)
        ref = orc.gcn_grouped(ps, tg, ix, vs, x, V, seg=0)
        y, y2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, F), 7.0, device=DEV)
        for kw in ({}, {"reduce": "mean"}, {"relu": True}):
            ds.run(dev(x), y, 128, "balanced", **kw)
            st.run(dev(x), y2, 128, "balanced", **kw)
            assert torch.equal(y, y2), kw
        ds.run(dev(x), y, 128, "balanced")
        assert np.array_equal(y.cpu().numpy(), ref)
        ds.run(dev(x), y, 128, "balanced", reduce="max")      # max stays on the streaming form
        assert np.array_equal(y.cpu().numpy(), orc.gcn_max(ptr, idx, v, x))
        if v is not None:                                       # the edge values follow the caller's array (updateval)
            v2 = rand(E, 9)
            ds.updateval(dev(v2))
            ds.run(dev(x), y, 128, "balanced")
            assert np.array_equal(y.cpu().numpy(), orc.gcn_grouped(ps, tg, ix, blocked_reference(ds, ptr, idx, v2)[3], x, V, seg=0))


@pytest.mark.parametrize("n", [1, 2, 3, 7])
def test_overlap_combine_chunks_keep_every_bit(n):
    """Option "overlap_combine" = 1 (one launch per column tile) / N >= 2 (N launches of consecutive tiles), the ordered combine of a
    chunk on the auxiliary stream beside the next chunk's aggregation: the same groups and folds as the single launch, bit for bit,
    GCN (sum / mean / max, 10 tiles of which the last is ragged) and GAT (4 tiles)."""
    V, E = 900, 260000
    ptr, idx = hub_graph(V, E, seed=5)
    F = 602
    x, val = rand((V, F), 1), rand(E, 2)
    a = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    b = gnc.Aggregator_GCN(dev(ptr), dev(idx), dev(val), F, F)
    for h in (a, b):
        h.set_option("slice_kb", 16)
    b.set_option("overlap_combine", n)
    assert a.balanced_partitions() == b.balanced_partitions() > 1
    y, y2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, F), 7.0, device=DEV)
    for kw in ({}, {"reduce": "mean"}, {"reduce": "max"}, {"relu": True}):
        for _ in range(2):   # the second call reuses the stream and the events
            a.run(dev(x), y, 128, "balanced", **kw)
            b.run(dev(x), y2, 128, "balanced", **kw)
            torch.cuda.synchronize()
            assert torch.equal(y, y2), (n, kw)
    F, H = 256, 8
    x, att = rand((V, F), 5), rand((V, H, 2), 6) * 0.4
    ga = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    gb = gnc.Aggregator_GAT(dev(ptr), dev(idx), F, F)
    for h in (ga, gb):
        h.set_option("slice_kb", 16)
    gb.set_option("overlap_combine", n)
    y, y2 = torch.full((V, F), 7.0, device=DEV), torch.full((V, F), 7.0, device=DEV)
    ga.run(dev(x), dev(att), y, 128, "balanced", heads=H)
    gb.run(dev(x), dev(att), y2, 128, "balanced", heads=H)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


