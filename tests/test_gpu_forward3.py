"""GPU: the 3-layer GCN / GAT forward of examples/forward_3layer.py (the harness of reference Figure7/our.py:171-188:
mm -> gcn_run(..., 128, 1) -> relu, and mm -> mm -> gat_run, three times, 512 -> 128 -> 64 -> 32) checked LAYER BY LAYER
against the oracle: every dense stage against orc_matmul_nn and every aggregation against orc_gcn_grouped_seg /
orc_gat_grouped_seg fed with the very tensors the GPU stage consumed, so an error cannot hide behind the next layer."""
import os
import sys

import numpy as np
import pytest
import torch

import gnn_computing_amd as gnc
from oracle import oracle as orc
from test_gpu_parity import DEV, assert_within, gat_scale

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import forward_3layer as f3  # noqa: E402

pytestmark = pytest.mark.gpu
NG = 32   # our.py:84


def graph(V=5000, E=70000):
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=123, alpha=0.9)     # a few rows with > 1000 edges: segments and hubs
    return ptr_t, idx_t


def order_of(agg, sched, ptr):
    """(ptr_s, target, seg) of the summation order the run used, from what the library reports."""
    if sched == 1:   # default ("fast_scheduled" = 1): scheduled = 1 runs the balanced order; the user's groups keep describing the schedule
        assert agg.mode_params("scheduled")[0] == NG
    chunk, seg = agg.balanced_params()
    ps, tg = orc.neighbor_grouping(ptr, chunk)
    return ps, tg, seg


def check_dense(a, b, c, exact, what):
    a, b, c = a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()
    ref = orc.matmul_nn(a, b)
    if exact:       # the library's f32-MFMA GEMM keeps the oracle's ascending-k chain
        assert np.array_equal(c, ref), what
    else:           # torch.mm (rocBLAS / hipBLASLt), as the reference script uses: another association of the same sum
        assert_within(c, ref, np.abs(a) @ np.abs(b), what)


@pytest.mark.parametrize("sched", [1, "balanced"])
@pytest.mark.parametrize("dense_name,fused_relu", [("torch.mm", False), ("matmul_NN", False), ("matmul_NN", True)])
def test_gcn_forward_layer_by_layer(sched, dense_name, fused_relu):
    ptr_t, idx_t = graph()
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    V = len(ptr) - 1
    dense = torch.mm if dense_name == "torch.mm" else gnc.matmul_NN
    m = f3.Model(ptr_t.to(DEV), idx_t.to(DEV), NG, sched, fused_relu, dense)
    m.trace = []
    y = m.forward("our_GCN")
    assert len(m.trace) == 3 and y.shape == (V, 32)
    ps, tg, seg = order_of(m.at, sched, ptr)
    ones = np.ones(len(idx), np.float32)
    prev = m.h
    for k, t in enumerate(m.trace):
        assert t["feat"] is prev or torch.equal(t["feat"], prev)                      # the layers are chained
        check_dense(t["feat"], t["w"], t["feat2"], dense_name != "torch.mm", "gcn layer %d dense" % k)
        ref = np.maximum(orc.gcn_grouped(ps, tg, idx, ones, t["feat2"].cpu().numpy(), V, seg=seg), 0)
        assert np.array_equal(t["out"].cpu().numpy(), ref), "gcn layer %d aggregation + relu" % k
        prev = t["out"]
    assert torch.equal(y, m.trace[-1]["out"])
    if dense_name != "torch.mm":   # every stage is bit-exact, so the whole forward is: restate it end to end on the CPU
        h = m.h.cpu().numpy()
        for k in range(3):
            h = np.maximum(orc.gcn_grouped(ps, tg, idx, ones, orc.matmul_nn(h, m.weights[k].cpu().numpy()), V, seg=seg), 0)
        assert np.array_equal(y.cpu().numpy(), h)


@pytest.mark.parametrize("sched", [1, "balanced"])
@pytest.mark.parametrize("dense_name", ["torch.mm", "matmul_NN"])
def test_gat_forward_layer_by_layer(sched, dense_name):
    ptr_t, idx_t = graph()
    ptr, idx = ptr_t.numpy(), idx_t.numpy()
    V = len(ptr) - 1
    dense = torch.mm if dense_name == "torch.mm" else gnc.matmul_NN
    m = f3.Model(ptr_t.to(DEV), idx_t.to(DEV), NG, sched, False, dense)
    m.trace = []
    y = m.forward("our_GAT")
    assert len(m.trace) == 3 and y.shape == (V, 32) and bool(torch.isfinite(y).all())
    ps, tg, seg = order_of(m.at_gat, sched, ptr)
    prev = m.h
    for k, t in enumerate(m.trace):
        assert t["feat"] is prev or torch.equal(t["feat"], prev)
        exact = dense_name != "torch.mm"
        check_dense(t["feat"], t["w"], t["feat2"], exact, "gat layer %d dense" % k)
        check_dense(t["feat2"], t["w_lr"], t["att"], exact, "gat layer %d attention terms" % k)
        feat2, att = t["feat2"].cpu().numpy(), t["att"].cpu().numpy()
        ref, _, _ = orc.gat_grouped(ps, tg, idx, att, feat2, V, 1, seg=seg)
        # the kernel keeps the oracle's association; what is left is device expf vs libm: 1e-5 of the weighted magnitude
        assert_within(t["out"].cpu().numpy(), ref, gat_scale(ptr, idx, att, feat2, 1) + np.abs(ref), "gat layer %d aggregation" % k)
        assert np.all(t["out"].cpu().numpy()[np.diff(ptr) == 0] == 0)
        prev = t["out"]
