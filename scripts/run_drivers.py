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
    print("==", exe, " ".join(extra), "rc", r.returncode, "GNNAGG_FAST_ROWS=" + os.environ.get("GNNAGG_FAST_ROWS", "default"),
          "GNNAGG_FAST_SCHEDULED=" + os.environ.get("GNNAGG_FAST_SCHEDULED", "default"))
    for line in r.stderr.splitlines():
        if line.startswith("{"):
            print("  ", line[:260])
# the REFERENCE's own drivers (Figure8 / Figure9 / Figure10 main*.cu, hipify-perl, no hand edits) on include/compat + libgnnagg.so:
# what a drop-in caller gets with no environment variables (both fig9 variants <= 95 us on the headline input)
import re
ref = os.path.join(ROOT, "oracle", "_ref", "drivers")
os.makedirs(os.path.join(d, "results"), exist_ok=True)
for exe, extra in (("fig9_ref.out", ["--nei", "32"]), ("fig9_ref.out", ["--nei", "32", "--reorder", "_thres_0.2"]),
                   ("fig10a_ref.out", ["--nei", "32"]), ("fig10b_ref.out", ["--nei", "64", "--outfea", "32"] if F == "32" else None),
                   ("fig8_ref.out", ["--nei", "16"] if F == "32" else None)):
    if extra is None or not os.path.exists(os.path.join(ref, exe)):
        continue
    r = subprocess.run([os.path.join(ref, exe), "--dataset", "arxiv", "--datadir", d, "--feature-len", F] + extra,
                       capture_output=True, text=True, timeout=600, cwd=d)
    print("== reference driver", exe, " ".join(extra), "rc", r.returncode)
    for line in r.stderr.splitlines():
        m = re.search(r"\] (\S.*getDuration\S* = [0-9.eE+-]+|hkz_\w+ = [0-9.eE+-]+)", line)
        if m:
            print("  ", m.group(1))
