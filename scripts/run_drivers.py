#!/usr/bin/env python3
"""Runs the C++ drivers (reference Figure 8 / 9 / 10a / 10b call sequences on the compat class shim) on the arxiv-shaped
synthetic data set and prints their JSON lines.  usage: run_drivers.py [feature_len]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnn_computing_amd as gnc  # noqa: E402

F = sys.argv[1] if len(sys.argv) > 1 else "128"
d = tempfile.mkdtemp() + "/"
ptr, idx = gnc.graph.dataset("arxiv")
gnc.graph.write_graph_files(d, "arxiv", ptr.numpy(), idx.numpy(), text=False, dumps=True)
rows, _ = gnc.cluster_reorder(ptr.numpy(), idx.numpy(), order="cache_greedy", cluster_cap=1, cache_rows=8192)
gnc.graph.write_reorder_file(d, "arxiv", np.asarray(rows, np.int32))
for exe, extra in (("fig8.out", ["--nei", "16"]), ("fig9.out", ["--nei", "32"]), ("fig9.out", ["--nei", "32", "--reorder", "_thres_0.2"]),
                   ("fig10a.out", ["--nei", "32"]), ("fig10b.out", ["--nei", "64", "--outfea", "32"])):
    r = subprocess.run([os.path.join(ROOT, "drivers", exe), "--dataset", "arxiv", "--datadir", d, "--feature-len", F] + extra,
                       capture_output=True, text=True, timeout=600)
    print("==", exe, " ".join(extra), "rc", r.returncode, "GNNAGG_FAST_ROWS=" + os.environ.get("GNNAGG_FAST_ROWS", "0"))
    for line in r.stderr.splitlines():
        if line.startswith("{"):
            print("  ", line[:260])
