import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The native pieces are built once per session (no-op when up to date)."""
    import __graft_entry__ as ge
    so = os.path.join(ROOT, "gnn_computing_amd", "libgnnagg.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        ge.build()
