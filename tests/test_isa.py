"""Static checks of the device code (hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_flat_memory_instructions():
    """No kernel of the library loads or stores through a generic pointer: FLAT operations count on lgkmcnt too, so every wait for
    an LDS read behind them waits for global memory (what held km_write_c_kernel back until round 3; tools/isa_flat.py)."""
    spec = importlib.util.spec_from_file_location("isa_flat", os.path.join(ROOT, "tools", "isa_flat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = mod.listing()
    try:
        assert mod.flat_by_kernel(path) == {}
    finally:
        shutil.rmtree(os.path.dirname(path), ignore_errors=True)
