"""CPU: the C-ABI library loads, exports every symbol include/gnnagg.h declares, and refuses to
compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import gnn_computing_amd as gnc
from gnn_computing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(extras=False):
    """the functions include/gnnagg.h declares: outside its `#ifdef GNNAGG_EXTRAS` block (the shipped surface) or inside it (Section E)"""
    text = open(os.path.join(ROOT, "include", "gnnagg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef GNNAGG_EXTRAS(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef GNNAGG_EXTRAS.*?#endif", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", inside if extras else outside)
    return sorted(set(names))


def test_header_symbols_are_exported_and_typed():
    names = declared_symbols()
    assert len(names) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in gnnagg.h but not exported: %s" % missing
    assert sorted(_lib.SIGNATURES) == names  # the Python binding types exactly the declared surface
    L = gnc.lib()
    assert L.gnnagg_version() >= 100
    # Section E (the backward entry points) is declared under GNNAGG_EXTRAS and exported by libgnnagg_extras.so ONLY
    extra = declared_symbols(extras=True)
    assert extra == sorted(_lib.EXTRA_SIGNATURES) and all("bwd" in n for n in extra)
    if _lib.has_extras():
        assert all(n in exported for n in extra)
    else:
        assert not [n for n in exported if "bwd" in n], "the shipped library must not export backward entry points"


def test_shipped_option_table_is_short():
    """gnnagg.h documents at most 12 per-handle options for the shipped library; the older forms are not among them"""
    text = open(os.path.join(ROOT, "include", "gnnagg.h")).read()
    table = text[text.index("/* Per-handle knobs."):text.index("int gnnagg_set_option(")]
    opts = re.findall(r'^ \*   "([a-z_]+)"', table, flags=re.M)
    assert len(opts) == len(set(opts)) <= 12, opts
    assert not set(opts) & {"retile", "tiled", "spans", "inkernel_combine", "host_plan", "partition_min_degree"}


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "gnnagg.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code and "Tensor" not in code


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_compute_refuses_without_gpu():
    L = gnc.lib()
    ptr = np.array([0, 1], np.int32)
    idx = np.array([0], np.int32)
    h = ctypes.c_int64(0)
    rc = L.gnnagg_gcn_create(ptr.ctypes.data, idx.ctypes.data, None, 1, 1, ctypes.byref(h))
    assert rc == _lib.ERR_HIP and h.value == 0
    assert b"no CPU fallback" in L.gnnagg_last_error()
    # flat API with abort disabled records the error and returns 0
    assert L.GCN_init_impl(ptr.ctypes.data, idx.ctypes.data, None, 1, 1) == 0
    with pytest.raises(ValueError):
        gnc.Aggregator_GCN(torch.from_numpy(ptr), torch.from_numpy(idx), None)  # CPU tensors are rejected


def test_bad_handle_and_arguments():
    L = gnc.lib()
    assert L.gnnagg_destroy(ctypes.c_int64(12345)) == _lib.ERR_ARG
    assert L.gnnagg_gcn_run(ctypes.c_int64(0), None, None, 4, 0, 0) == _lib.ERR_ARG
    n = ctypes.c_int(0)
    assert L.gnnagg_neighbor_grouping_schedule(None, 4, 3, None, None, ctypes.byref(n)) == _lib.ERR_ARG
    assert L.gnnagg_partition_rows(None, 3, 2, None) == _lib.ERR_ARG
    # round 5: the gather-ceiling probe checks its geometry before it touches the device; the transport query rejects a dead communicator
    buf = np.zeros(1024, np.int32)
    for pitch, seg, n_ids, per_group in [(16, 16, 0, 8), (24, 16, 256, 8), (512, 520, 512, 64), (512, 2048, 512, 64), (512, 512, 100, 32), (512, 512, 512, 24)]:
        assert L.gnnagg_probe_row_gather(buf.ctypes.data, pitch, seg, buf.ctypes.data, n_ids, per_group, None) == _lib.ERR_ARG, (pitch, seg, n_ids, per_group)
    assert L.gnnagg_probe_row_gather(None, 512, 512, buf.ctypes.data, 512, 32, None) == _lib.ERR_ARG
    path, bus, over = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64), ctypes.c_int(7)
    assert L.gnnagg_dist_transport_info(ctypes.c_int64(4242), path, 64, bus, 64, ctypes.byref(over)) == _lib.ERR_ARG


def test_file_rendezvous_rejects_what_a_crashed_launch_left_behind(tmp_path):
    """gnnagg_dist_comm_create_from_file: a rank > 0 must never take a stale id file -- ncclCommInitRank with
    mismatched ids hangs forever.  It accepts only a file that carries the token it published itself; anything else at the path
    (an id file of the old 128-byte format, a well-formed record of another launch) is ignored until the wait times out, and the
    request file is cleaned up."""
    import struct
    import time
    L = gnc.lib()
    path = str(tmp_path / "gnnagg.id")
    out = ctypes.c_int64(0)
    for stale in (b"\x07" * 128, struct.pack("<QQ", 0x31444947414E4E47, 2) + b"\x01" * 128 + struct.pack("<Q", 0xDEADBEEF)):
        open(path, "wb").write(stale)
        t0 = time.time()
        rc = L.gnnagg_dist_comm_create_from_file(path.encode(), 1, 2, 1, ctypes.byref(out))
        assert rc == _lib.ERR_IO and out.value == 0 and b"timed out" in L.gnnagg_last_error()
        assert 0.9 < time.time() - t0 < 10
        assert not os.path.exists(path + ".req.1")
    assert L.gnnagg_dist_comm_create_from_file(path.encode(), 2, 2, 1, ctypes.byref(out)) == _lib.ERR_ARG
