"""CPU suite: the C ABI at COMPILE level -- the db_builder patch and the multi-GPU call sequence of INTEGRATION.md are
extracted from the document, compiled with g++ -std=c++17 against include/ipkgpu.h (with a small mock of the db_builder
members they touch) and linked with libipkgpu.so; ctypes alone would only check symbol names."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "db_builder::explore_kmers()" in b]
    multi = [b for b in blocks if "ipkgpu_exchange_begin" in b and "build_shard" in b]
    assert len(stub) == 1 and len(multi) == 1, "INTEGRATION.md must hold the db_builder patch and the multi-GPU sequence"
    (tmp_path / "integration_stub.inc").write_text(stub[0])
    (tmp_path / "integration_multi.inc").write_text(multi[0])
    exe = tmp_path / "abi_stub"
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror=implicit-function-declaration", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
           "-I", os.path.join(ROOT, "tests", "abi_stub"), os.path.join(ROOT, "tests", "abi_stub", "main.cpp"), "-o", str(exe),
           "-L", os.path.join(ROOT, "ipk_amd"), "-lipkgpu", "-Wl,-rpath," + os.path.join(ROOT, "ipk_amd"), "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def test_integration_stubs_compile_link_and_refuse_without_a_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0, (r.stdout, r.stderr)
    else:
        assert r.returncode == 3 and "no CPU fallback" in r.stderr, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_integration_stub_runs_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "groups 3" in r.stdout, (r.stdout, r.stderr)
