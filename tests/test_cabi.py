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


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnnagg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
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
