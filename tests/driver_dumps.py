"""What a C++ driver COMPUTED, checked from outside: with GNNAGG_COMPAT_DUMP=<dir> (drivers/: `--dump <dir>`) the class
shim of include/compat/ writes the operands of the last call of each entry point as raw arrays; these helpers read them and compare them
with the oracle at the suite's bounds -- bit-exact where the kernel keeps the restated order, |y - ref| <= 1e-5 * sum|v x| elsewhere.
Used for the drivers under drivers/ (tests/test_gpu_parity.py) and for the reference's own Figure9 / Figure10 drivers built against the
shim (tests/test_gpu_reference.py): the class boundary is tested for numbers, not for liveness."""
import os

import numpy as np

from oracle import oracle as orc


def read(d, entry, operand, dtype=np.float32):
    path = os.path.join(d, "%s.%s.bin" % (entry, operand))
    assert os.path.exists(path), "the driver did not dump %s.%s (entry point never called?)" % (entry, operand)
    return np.fromfile(path, dtype=dtype)


def graph(d, entry, ptr, idx):
    """the CSR the aggregator ran on must be the one the test wrote (after load_graph's reorder): bit-exact"""
    p, i = read(d, entry, "ptr", np.int32), read(d, entry, "idx", np.int32)
    assert np.array_equal(p, ptr) and np.array_equal(i, idx), entry + ": the driver's CSR differs from the expected (reordered) CSR"
    return p, i


def gcn(d, entry, ptr, idx, F, exact=False, use_val=True):
    p, i = graph(d, entry, ptr, idx)
    V = len(p) - 1
    x, y = read(d, entry, "x").reshape(V, F), read(d, entry, "y").reshape(V, F)
    val = read(d, entry, "val") if use_val else None
    ref = orc.gcn_seq(p, i, val, x)
    if exact:
        assert np.array_equal(y, ref), entry + ": not bit-equal to the CSR-order chain (aggr_gcn.h:13-35)"
    else:
        scale = orc.gcn_abs_scale(p, i, val, x)
        assert np.all(np.abs(y - ref) <= 1e-5 * scale + 1e-30), entry + ": outside 1e-5 * sum|v x| of the CSR-order chain"
    return x, val, y


def gat_scale(ptr, idx, att, x, heads):
    V, F = x.shape
    wn = orc.gat_att(ptr, idx, att, heads, 0.2)
    sc = np.zeros((V, F))
    np.add.at(sc, np.repeat(np.arange(V), np.diff(ptr)), np.repeat(wn, F // heads, axis=1).astype(np.float64) * np.abs(x[idx]))
    return sc


def gat(d, entry, ptr, idx, F, heads=1):
    p, i = graph(d, entry, ptr, idx)
    V = len(p) - 1
    x, y = read(d, entry, "x").reshape(V, F), read(d, entry, "y").reshape(V, F)
    att = read(d, entry, "att").reshape(V, heads, 2)
    ref = orc.gat_fused(p, i, att, x, heads)
    nz = np.diff(p) > 0                                  # (empty rows: 0 here, NaN in the reference's aggr_gat -- SURVEY 8a)
    assert np.all(y[~nz] == 0)
    bound = 1e-5 * (gat_scale(p, i, att, x, heads) + np.abs(np.nan_to_num(ref))) + 1e-30
    assert np.all(np.abs(y[nz] - ref[nz]) <= bound[nz]), entry + ": outside the 1e-5 bound of the fused edge softmax (aggr_gat.h:116-164)"


def edge_softmax_stages(d, ptr, idx, with_exp):
    """the unfused stages of Figure10/main_a.cu's base variant (aggr_gat.h:33-92) and run_att (aggr_gat.h:5-31), each against the oracle on
    the inputs the driver gave it"""
    V = len(ptr) - 1
    graph(d, "gat_run_u_add_v", ptr, idx)
    att = read(d, "gat_run_u_add_v", "att").reshape(V, 2)
    # the last u_add_v output was overwritten by the later stages; its inputs and the oracle give it back, exactly
    s = orc.gat_u_add_v(ptr, idx, att)
    val_in = read(d, "gat_run_add_to_center", "val_in")
    if not with_exp:                                     # the reference's driver feeds u_add_v's sums straight into add_to_center
        assert np.array_equal(val_in, s), "run_u_add_v: not exact"
    else:                                                # drivers/fig10a.cpp: exp(leaky_relu) between them, like our.py:145-151
        np.testing.assert_allclose(val_in, np.exp(np.maximum(s, np.float32(0.2) * s)), rtol=1e-5, err_msg="run_u_add_v + exp")
    center = read(d, "gat_run_add_to_center", "att")
    ref_c = orc.gat_add_to_center(ptr, val_in)
    seg_abs = orc.gat_add_to_center(ptr, np.abs(val_in))
    assert np.all(np.abs(center - ref_c) <= 1e-5 * seg_abs + 1e-30), "run_add_to_center outside 1e-5 * sum|v|"
    val_d = read(d, "gat_run_div_each", "val_in")
    out = read(d, "gat_run_div_each", "val")
    ref_d = orc.gat_div_each(ptr, read(d, "gat_run_div_each", "att"), val_d)
    fin = np.isfinite(ref_d) & np.isfinite(out)
    assert np.array_equal(np.isfinite(ref_d), np.isfinite(out)) and np.allclose(out[fin], ref_d[fin], rtol=2e-6, atol=0), "run_div_each"
    a2 = read(d, "gat_run_att", "att").reshape(V, 1, 2)
    w = read(d, "gat_run_att", "val")
    np.testing.assert_allclose(w, orc.gat_att(ptr, idx, a2, 1, 0.2)[:, 0], rtol=1e-5, err_msg="run_att")


def run_with_nn(d, ptr, idx, F, OUT):
    """aggr_gcn.h:491-499: vout = A.vin inside the bound, transformed = vout . W bit-equal to the ascending-k chain ON THAT vout"""
    _, _, y = gcn(d, "gcn_run_with_nn", ptr, idx, F)
    w = read(d, "gcn_run_with_nn", "weight").reshape(F, OUT)
    t = read(d, "gcn_run_with_nn", "transformed").reshape(len(ptr) - 1, OUT)
    assert np.array_equal(t, orc.matmul_nn(y, w)), "run_with_nn: transformed is not the ascending-k chain of vout . weight"


def matmul(d, entry="matmul_NN", M=None, K=None, N=None):
    a, b, c = read(d, entry, "A").reshape(M, K), read(d, entry, "B").reshape(K, N), read(d, entry, "C").reshape(M, N)
    assert np.array_equal(c, orc.matmul_nn(a, b)), "matmul_NN: not bit-equal to the ascending-k fmaf chain (dense.h:4-23)"
