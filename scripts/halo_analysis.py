#!/usr/bin/env python3
"""Projected exchange volume of the row-partitioned step, from the plans alone (host only, no GPU): per rank the halo rows a
PULL moves (distinct remote source rows, dedup'ed per owner) against what a PUSH would move for the same peer pair (partial Y
rows: distinct (destination row, owner) pairs), and the best of the two chosen per peer pair.
usage: halo_analysis.py A 2 4 8   |   halo_analysis.py P 8        (A: N x arxiv-shaped weak scaling, F = 128; P: products-shaped, F = 100)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

cfg = sys.argv[1]
worlds = [int(a) for a in sys.argv[2:]] or [8]
for world in worlds:
    V1, E1 = gnc.graph.SHAPES["products" if cfg == "P" else "arxiv"]
    V, E = (V1, E1) if cfg == "P" else (V1 * world, E1 * world)
    F = 100 if cfg == "P" else 128
    ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=123, community_order=True)
    ptr, idx = ptr_t.numpy().astype(np.int64), idx_t.numpy().astype(np.int64)
    bounds = gnc.partition_rows(ptr.astype(np.int32), world).astype(np.int64)
    owner_of = np.searchsorted(bounds, np.arange(V), side="right") - 1
    rows = np.repeat(np.arange(V), np.diff(ptr))
    dst_rank, src_rank = owner_of[rows], owner_of[idx]
    remote = dst_rank != src_rank
    pull = np.zeros((world, world), np.int64)   # [reader, owner]: distinct source rows
    push = np.zeros((world, world), np.int64)   # [reader, owner]: distinct destination rows with a source at `owner`
    key_pull = np.unique(dst_rank[remote] * V + idx[remote])
    np.add.at(pull, (key_pull // V, owner_of[key_pull % V]), 1)
    key_push = np.unique(src_rank[remote] * V + rows[remote])
    np.add.at(push, (owner_of[key_push % V], key_push // V), 1)
    best = np.minimum(pull, push)
    per_rank = lambda m: float(m.sum(1).mean())  # noqa: E731
    print(json.dumps({
        "config": cfg, "world": world, "feat": F, "remote_edge_share": float(remote.mean()),
        "pull_rows_per_rank": per_rank(pull), "push_rows_per_rank": per_rank(push), "best_of_both_per_pair_rows_per_rank": per_rank(best),
        "pull_MB_per_rank_per_step": per_rank(pull) * F * 4 / 1e6, "push_MB_per_rank_per_step": per_rank(push) * F * 4 / 1e6,
        "best_MB_per_rank_per_step": per_rank(best) * F * 4 / 1e6,
        "max_pair_MB_pull": float(pull.max()) * F * 4 / 1e6, "own_x_MB_per_rank": V / world * F * 4 / 1e6}), flush=True)
