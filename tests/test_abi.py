"""CPU suite: the C-ABI library loads and exports exactly what include/ipkgpu.h declares."""
import ctypes
import os
import re

import pytest

import ipk_amd
from ipk_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ipkgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ipkgpu_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "ipk_amd", "libipkgpu.so"))
    declared = _declared()
    assert declared, "header parse failed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ipkgpu.h but not exported"
    from ipk_amd import dbfile, loader, tree
    assert sorted(E.ABI_SYMBOLS + loader.ABI_SYMBOLS + tree.ABI_SYMBOLS + dbfile.ABI_SYMBOLS) == declared


def test_host_helpers_no_gpu_needed():
    assert abs(ipk_amd.log_threshold(1.5, 4, 10) - (-4.259687)) < 1e-5
    assert ipk_amd.bits_per_symbol(4) == 2 and ipk_amd.bits_per_symbol(20) == 5 and ipk_amd.bits_per_symbol(7) == 0
    assert ipk_amd.kmer_batch(1000003, 32) == 1000003 % 32
    assert ipk_amd.max_k(4) == 14 and ipk_amd.max_k(20) == 6


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never fall back to CPU code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ipk_amd.IpkGpuError) as ei:
        ipk_amd.Engine(0)
    assert ei.value.code == 4
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_use_oracle():
    """The oracle is test infrastructure: nothing under ipk_amd/ may import, load or link it."""
    bad = ("import oracle", "from oracle", "libipk_oracle", "ipk_oracle", "ipko_")
    for root, _, files in os.walk(os.path.join(ROOT, "ipk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                for b in bad:
                    assert b not in text, f"ipk_amd/{f} references the oracle ({b})"
