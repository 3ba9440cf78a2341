import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suites load the in-tree C-ABI library; build it (hipcc cross-compiles gfx950 without a GPU) if a fresh
    # checkout has not run __graft_entry__.build() yet
    from ipk_amd import build as hip_build
    hip_build.build()


@pytest.fixture(scope="session")
def engine():
    """One scoring context for the whole GPU session (all GPU tests run in one process)."""
    import ipk_amd
    eng = ipk_amd.Engine(0)
    yield eng
    eng.close()
