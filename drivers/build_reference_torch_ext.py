#!/usr/bin/env python3
"""drivers/build_reference_torch_ext.py -- the drop-in claim at the PyTorch boundary, exercised: the REFERENCE's own torch
binding (/root/reference/Figure7/kernel.cpp: pybind wrappers over ten flat functions + new_load), translated where it lies by
ROCm's hipify-perl (no hand edits) into a scratch directory whose ../include is THIS repo's include/compat, compiled with
torch.utils.cpp_extension together with drivers/flat_cxx_linkage.cpp and linked with libgnnagg.so.  Output:
oracle/_ref/drivers/gnncompile.so (git-ignored; tests/test_gpu_reference.py imports it and runs the reference's Python-level call
sequence, Figure7/our.py:171-188).  Skips quietly when the reference tree or hipify-perl is missing."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REF", "/root/reference")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
OUT = os.path.join(ROOT, "oracle", "_ref", "drivers", "gnncompile.so")


def up_to_date():
    if not os.path.exists(OUT) or os.environ.get("FORCE"):
        return False
    deps = [os.path.join(REF, "Figure7", "kernel.cpp"), os.path.join(HERE, "flat_cxx_linkage.cpp"), os.path.abspath(__file__),
            os.path.join(ROOT, "include", "gnnagg.h")] + glob.glob(os.path.join(ROOT, "include", "compat", "*.h"))
    return all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps)


def main():
    hipify = os.path.join(ROCM, "bin", "hipify-perl")
    src = os.path.join(REF, "Figure7", "kernel.cpp")
    if not os.path.exists(src) or not os.path.exists(hipify):
        print("oracle/_ref/drivers/gnncompile.so: no reference tree: not built")
        return
    if up_to_date():
        print("oracle/_ref/drivers/gnncompile.so is up to date")
        return
    from torch.utils.cpp_extension import load
    gen = tempfile.mkdtemp(prefix="gnnreftorch.")
    try:
        os.makedirs(os.path.join(gen, "Figure7"))
        os.symlink(os.path.join(ROOT, "include", "compat"), os.path.join(gen, "include"))   # kernel.cpp includes "../include/util.h"
        with open(os.path.join(gen, "Figure7", "kernel.cpp"), "w") as f:
            subprocess.check_call([hipify, src], stdout=f, stderr=subprocess.DEVNULL)
        load(name="gnncompile", sources=[os.path.join(gen, "Figure7", "kernel.cpp"), os.path.join(HERE, "flat_cxx_linkage.cpp")],
             extra_include_paths=[os.path.join(ROCM, "include"), os.path.join(ROCM, "include", "hiprand"), os.path.join(ROCM, "include", "hipblas"),
                                  os.path.join(ROOT, "include")],
             extra_cflags=["-D__HIP_PLATFORM_AMD__", "-w", "-std=c++17"],
             extra_ldflags=["-L" + os.path.join(ROOT, "gnn_computing_amd"), "-lgnnagg", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lhiprand",
                            "-lhipblas", "-Wl,-rpath," + os.path.join(ROOT, "gnn_computing_amd"), "-Wl,-rpath," + os.path.join(ROCM, "lib")],
             build_directory=gen, verbose=False, is_python_module=False)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        shutil.copy(os.path.join(gen, "gnncompile.so"), OUT)
        print("oracle/_ref/drivers/gnncompile.so: the reference's Figure7/kernel.cpp built against include/compat + libgnnagg.so")
    finally:
        shutil.rmtree(gen, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
