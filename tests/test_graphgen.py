"""CPU: the synthetic input generator (shapes, determinism, structure)."""
import numpy as np

from gnn_computing_amd import graph


def test_powerlaw_shape_and_structure():
    V, E = 5000, 60000
    ptr, idx = graph.powerlaw_csr(V, E, seed=123)
    ptr, idx = ptr.numpy(), idx.numpy()
    assert ptr[0] == 0 and ptr[-1] == E and len(ptr) == V + 1 and len(idx) == E
    deg = np.diff(ptr)
    assert (deg >= 0).all() and (deg == 0).sum() >= 1 and deg.max() > 20 * deg.mean()  # hubs + empty rows
    assert idx.min() >= 0 and idx.max() < V
    for r in np.random.default_rng(0).integers(0, V, 200):  # neighbors sorted inside a row
        assert np.all(np.diff(idx[ptr[r]:ptr[r + 1]]) >= 0)
    p2, i2 = graph.powerlaw_csr(V, E, seed=123)
    assert np.array_equal(p2.numpy(), ptr) and np.array_equal(i2.numpy(), idx)  # deterministic
    p3, i3 = graph.powerlaw_csr(V, E, seed=124)
    assert not np.array_equal(i3.numpy(), idx)


def test_arxiv_shape_exact():
    ptr, idx = graph.dataset("arxiv")
    assert (len(ptr) - 1, len(idx)) == graph.SHAPES["arxiv"] == (169343, 1166243)
    assert int(ptr[-1]) == 1166243


def test_community_order_is_more_local():
    V, E = 20000, 200000
    a = graph.powerlaw_csr(V, E, seed=1, community_order=False)
    b = graph.powerlaw_csr(V, E, seed=1, community_order=True)
    def spread(ptr, idx):
        rows = np.repeat(np.arange(V), np.diff(ptr.numpy()))
        return np.median(np.abs(idx.numpy().astype(np.int64) - rows))
    assert spread(*b) < 0.5 * spread(*a)
