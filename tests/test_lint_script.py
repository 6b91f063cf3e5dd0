"""CPU: scripts/check_store_hazard.py (the ISA lint behind `make -C gnn_computing_amd/csrc lint`) must FAIL on what ADVICE r5 found it passing:
an assembly file that did not compile (no kernel in it) and a required kernel that is missing -- a stale or partial .s file may never read
as "0 suspicious place(s)"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "check_store_hazard.py")

GOOD = """\t.text
\t.amdhsa_kernel _ZN6gnnagg16k_dense_nn_aheadEPKfS1_Pfiiiii
_ZN6gnnagg16k_dense_nn_aheadEPKfS1_Pfiiiii:
\tv_mov_b32 v1, 0
\ts_endpgm
.Lfunc_end0:
"""
HAZARD = """\t.text
\t.amdhsa_kernel _Z1kv
_Z1kv:
\tbuffer_store_dwordx4 v[4:7], v0, s[0:3], 0 offen
.LBB0_1:
\tv_mov_b32 v5, 0
\ts_endpgm
.Lfunc_end0:
"""


def run(args):
    return subprocess.run([sys.executable, SCRIPT] + args, capture_output=True, text=True, timeout=60)


def test_lint_passes_clean_assembly_and_finds_the_hazard(tmp_path):
    good, bad = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(GOOD)
    bad.write_text(HAZARD)
    r = run(["--require", "k_dense_nn_ahead", str(good)])
    assert r.returncode == 0 and "0 suspicious place(s)" in r.stdout and "0 violation(s)" in r.stdout, r.stdout + r.stderr
    r = run([str(bad)])
    assert r.returncode != 0 and "1 suspicious place(s)" in r.stdout


def test_lint_fails_on_a_file_that_did_not_compile_or_lacks_a_required_kernel(tmp_path):
    empty, good = tmp_path / "empty.s", tmp_path / "good.s"
    empty.write_text("")            # what a failed hipcc leaves behind
    good.write_text(GOOD)
    r = run([str(good), str(empty)])
    assert r.returncode != 0 and "did not compile" in (r.stdout + r.stderr)
    r = run(["--require", "k_dense_nn_ahead,k_dense_nn_ahead2", str(good)])
    assert r.returncode != 0 and "k_dense_nn_ahead2" in (r.stdout + r.stderr)
    assert run([]).returncode != 0
