#!/usr/bin/env python3
"""Wall time and window footprint of the reorder generator on the products-shaped graph, on this host's cores (GNNAGG_LIB picks a build)."""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_reorder_generator import _window_footprint  # noqa: E402

p, i = gnc.graph.dataset("products")
ptr, idx = p.numpy(), i.numpy()
V = len(ptr) - 1
t = time.time()
o, _ = gnc.cluster_reorder(ptr, idx, order="cache_greedy", cluster_cap=1, cache_rows=8192)
dt = time.time() - t
f = _window_footprint(ptr.astype(np.int64), idx.astype(np.int64), o.astype(np.int64))
print("%s: %.1f s, window footprint %.4f, valid %s, sha %s, threads %s" % (os.environ.get("GNNAGG_LIB", "main").split("_")[-1], dt, f,
      np.array_equal(np.sort(o), np.arange(V)), hashlib.sha256(o.tobytes()).hexdigest()[:16], os.cpu_count()), flush=True)
