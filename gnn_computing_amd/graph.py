"""Synthetic CSR inputs at the reference's dataset shapes, and writers for its file formats.

The reference's datasets are an external download (README.md:45-54) that is not available; the
benchmark and the full-size tests therefore use a seeded (123 -- the reference's seed,
Figure9/main.cu:22) power-law generator with the exact (|V|,|E|) of util.py:25-33,110-123.

Generator: in-degrees follow (rank+1)^-alpha over a seeded permutation of
the nodes (hubs scattered), rescaled to sum exactly to E; a fraction p_local of each row's
sources is drawn from a window around the row's position in a hidden community order (this is
the structure a locality reorder can recover), the rest from the same power-law popularity
(hub sources); neighbors are sorted inside a row; duplicates are allowed; >= 1 row is empty.
"""
import os

import numpy as np
import torch

SHAPES = {
    # name: (V, E)  -- CSR nnz, directed (reference util.py:25,31,33,110,120,123)
    "arxiv": (169343, 1166243),
    "reddit": (232965, 114615891),
    "products": (2449029, 123718280),
}


def _degrees(V, E, alpha, gen, device):
    rank = torch.randperm(V, generator=gen, device=device).to(torch.float64)
    w = (rank + 1.0).pow(-alpha)
    target = w * (E / w.sum())
    deg = target.floor().clamp_(max=V - 1)
    # force the lightest rows empty (real graphs have isolated / sink nodes; exercises the
    # empty-row path at every size)
    n_empty = max(1, V // 1000)
    order = torch.argsort(w)
    deg[order[:n_empty]] = 0
    # hand the remaining edges to the rows with the largest fractional part (ties by weight)
    rem = int(E - int(deg.sum().item()))
    while rem > 0:
        room = (deg < V - 1)
        room[order[:n_empty]] = False
        frac = torch.where(room, target - deg, torch.full_like(target, -1e30))
        k = min(rem, int(room.sum().item()))
        top = torch.topk(frac, k).indices
        deg[top] += 1
        rem -= k
    assert rem == 0
    return deg.to(torch.int64), w


def powerlaw_csr(V, E, seed=123, alpha=0.75, p_local=0.5, window=1024, device="cpu", community_order=False):
    """Returns (ptr[V+1] int32, idx[E] int32) torch tensors on `device`.
    community_order=True numbers the nodes in the hidden community order (the graph a perfect
    locality reorder would produce); False scatters them with a seeded permutation."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if V <= 1 or E == 0:
        return (torch.zeros(V + 1, dtype=torch.int32, device=device), torch.zeros(0, dtype=torch.int32, device=device))
    deg, w = _degrees(V, E, alpha, gen, device)
    ptr = torch.zeros(V + 1, dtype=torch.int64, device=device)
    ptr[1:] = torch.cumsum(deg, 0)
    rows = torch.repeat_interleave(torch.arange(V, device=device), deg)
    # hidden community order: sigma[node] = position
    sigma = torch.randperm(V, generator=gen, device=device)
    if community_order:
        sigma = torch.arange(V, device=device)
    inv_sigma = torch.empty_like(sigma)
    inv_sigma[sigma] = torch.arange(V, device=device)
    # global (popularity) picks through the inverse CDF of the same power law
    cdf = torch.cumsum(w / w.sum(), 0)
    u = torch.rand(E, generator=gen, device=device, dtype=torch.float64)
    glob = torch.searchsorted(cdf, u).clamp_(max=V - 1)
    # local picks: window around the row's community position
    off = torch.randint(-window, window + 1, (E,), generator=gen, device=device)
    loc = inv_sigma[(sigma[rows] + off) % V]
    pick_local = torch.rand(E, generator=gen, device=device) < p_local
    src = torch.where(pick_local, loc, glob)
    # sort neighbors inside each row (real CSRs are sorted)
    key = rows * V + src
    key, _ = torch.sort(key)
    idx = (key % V).to(torch.int32)
    return ptr.to(torch.int32), idx


def dataset(name, device="cpu", seed=123):
    V, E = SHAPES[name]
    return powerlaw_csr(V, E, seed=seed, device=device)


def uniform_random_csr(V, E, seed=0):
    """Small numpy helper for tests: ragged random CSR with empty rows, unsorted neighbors."""
    rng = np.random.default_rng(seed)
    deg = rng.multinomial(E, rng.dirichlet(np.full(V, 0.3))) if V > 0 else np.zeros(0, np.int64)
    ptr = np.zeros(V + 1, np.int32)
    ptr[1:] = np.cumsum(deg)
    idx = rng.integers(0, max(V, 1), size=E).astype(np.int32)
    return ptr, idx


# ------------------------------------------------------------------- reference file formats
def write_graph_files(datadir, dset, ptr, idx, text=True, dumps=False):
    """<dset>.config ("V E"), <dset>.graph (line 1 ptr, line 2 idx; README.md:74-75) and/or the raw
    little-endian int32 caches <dset>.graph.ptrdump/.edgedump (data.cu:52-53,79-80)."""
    ptr = np.asarray(ptr, dtype=np.int32)
    idx = np.asarray(idx, dtype=np.int32)
    os.makedirs(datadir, exist_ok=True)
    base = os.path.join(datadir, dset)
    with open(base + ".config", "w") as f:
        f.write("%d %d" % (len(ptr) - 1, len(idx)))
    if text:
        with open(base + ".graph", "w") as f:
            f.write(" ".join(map(str, ptr.tolist())) + "\n")
            f.write(" ".join(map(str, idx.tolist())) + "\n")
    if dumps:
        ptr.astype("<i4").tofile(base + ".graph.ptrdump")
        idx.astype("<i4").tofile(base + ".graph.edgedump")


def write_reorder_file(datadir, dset, rows, suffix="_thres_0.2"):
    """<dset>.reorder<suffix>: V ints, entry i = old node id placed at new position i
    (data.cu:3,105-113; README.md:76)."""
    with open(os.path.join(datadir, dset + ".reorder" + suffix), "w") as f:
        f.write(" ".join(map(str, np.asarray(rows).tolist())))


def reorder_on_load(name, ptr, idx, key="x", cache_dir=None):
    """The locality reorder applied ON LOAD through the reference's own file formats (src/data.cu:96-133; our.py:79): writes
    <dset>.config + <dset>.graph.ptrdump / .edgedump (data.cu:52-53,79-80) and <dset>.reorder_thres_0.2 (V ints, entry i = old id at new
    position i, data.cu:105-113) into `cache_dir` (default: a per-user directory under the system temp dir), then lets the library's
    loader (gnnagg_load_graph) read them and call its reorderCSR.  The permutation is the library generator's (gnnagg_cluster_reorder_ex:
    cache-aware greedy order over singleton clusters, cache model 8192 rows); it is generated once per (name, key) and found again by
    later processes on the same box.  Returns (ptr, idx, rows, seconds generating [0.0 on a cache hit], seconds loading)."""
    import tempfile
    import time
    from .aggregator import cluster_reorder, load_graph_host
    d = (cache_dir or os.path.join(tempfile.gettempdir(), "gnnagg_bench_cache_%d" % os.getuid())) + os.sep
    dset = "%s_%s" % (name, key)
    os.makedirs(d, exist_ok=True)
    ptr, idx = np.ascontiguousarray(ptr, np.int32), np.ascontiguousarray(idx, np.int32)
    dump = d + dset + ".graph.edgedump"
    if not (os.path.exists(dump) and os.path.getsize(dump) == 4 * len(idx) and os.path.exists(d + dset + ".config")):
        write_graph_files(d, dset, ptr, idx, text=False, dumps=True)
    t_gen = 0.0
    if not os.path.exists(d + dset + ".reorder_thres_0.2"):
        t0 = time.perf_counter()
        rows, _ = cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
        t_gen = time.perf_counter() - t0
        tmp = dset + ".tmp%d" % os.getpid()
        write_reorder_file(d, tmp, rows)
        os.replace(d + tmp + ".reorder_thres_0.2", d + dset + ".reorder_thres_0.2")
    t0 = time.perf_counter()
    g = load_graph_host(dset, "_thres_0.2", d, shuffle=True)
    t_load = time.perf_counter() - t0
    if g["rows"] is None or g["num_v"] != len(ptr) - 1 or g["num_e"] != len(idx):
        raise RuntimeError("reorder_on_load: the loader did not apply %s%s.reorder_thres_0.2" % (d, dset))
    return g["ptr"], g["idx"], g["rows"], t_gen, t_load


def locality_order(ptr, idx):
    """Interim locality reorder (stand-in for the reference's offline MinHash-LSH clustering,
    script/cluster2.py -- a "next" row of SURVEY.md 8f): reverse Cuthill-McKee on the symmetrised
    graph.  Returns rows[] in the reorder-file convention (rows[i] = old id at new position i)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    ptr = np.asarray(ptr, dtype=np.int64)
    idx = np.asarray(idx, dtype=np.int64)
    V = len(ptr) - 1
    a = sp.csr_matrix((np.ones(len(idx), np.int8), idx, ptr), shape=(V, V))
    a = (a + a.T).tocsr()
    return np.asarray(reverse_cuthill_mckee(a, symmetric_mode=True), dtype=np.int32)
