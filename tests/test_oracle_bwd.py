"""The oracle's GAT backward (orc_gat_bwd, restating aggr_gat.h:222-296) against central finite differences of a float64
forward pass -- CPU only."""
import numpy as np

from oracle import oracle as orc


def _forward(ptr, idx, x, att, slope):
    V, F = x.shape
    E = len(idx)
    out, w, D = np.zeros((V, F)), np.zeros(E), np.zeros(V)
    for r in range(V):
        for e in range(ptr[r], ptr[r + 1]):
            z = att[r, 0] + att[idx[e], 1]
            w[e] = np.exp(z if z > 0 else slope * z)
            D[r] += w[e]
        for e in range(ptr[r], ptr[r + 1]):
            out[r] += w[e] / D[r] * x[idx[e]]
    return out, w, D


def test_gat_bwd_matches_finite_differences():
    rng = np.random.default_rng(0)
    V, F, slope = 9, 5, 0.2
    deg = np.array([3, 0, 3, 2, 1, 3, 1, 0, 4])
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    idx = rng.integers(0, V, int(ptr[-1])).astype(np.int32)
    x, att, G = rng.standard_normal((V, F)), rng.standard_normal((V, 2)) * 0.7, rng.standard_normal((V, F))
    out, w, D = _forward(ptr, idx, x, att, slope)
    f32 = lambda a: a.astype(np.float32)  # noqa: E731
    d_a_b, d_feat = orc.gat_bwd(ptr, idx, f32(out), f32(G), f32(w), f32(D), f32(x), slope)
    eps = 1e-6
    num_ab = np.zeros((V, 2))
    for i in range(V):
        for j in range(2):
            ap, am = att.copy(), att.copy()
            ap[i, j] += eps
            am[i, j] -= eps
            num_ab[i, j] = ((_forward(ptr, idx, x, ap, slope)[0] - _forward(ptr, idx, x, am, slope)[0]) * G).sum() / (2 * eps)
    num_x = np.zeros((V, F))
    for i in range(V):
        for j in range(F):
            xp, xm = x.copy(), x.copy()
            xp[i, j] += eps
            xm[i, j] -= eps
            num_x[i, j] = ((_forward(ptr, idx, xp, att, slope)[0] - _forward(ptr, idx, xm, att, slope)[0]) * G).sum() / (2 * eps)
    np.testing.assert_allclose(d_a_b, num_ab, atol=2e-6)
    np.testing.assert_allclose(d_feat, num_x, atol=2e-6)
    assert np.all(d_a_b[deg == 0, 0] == 0)
