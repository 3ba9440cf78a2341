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
    """The db_builder patch of INTEGRATION.md, compiled, fills the (mock) database on the GPU; what it inserted -- scored count and
    every (key, branch, score bits) triple -- is compared with the oracle run on the very same matrices (the stub dumps them)."""
    import numpy as np
    from oracle import ipk_oracle as co
    exe = _build(tmp_path)
    dump = tmp_path / "mats.f32"
    r = subprocess.run([str(exe), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "groups 3" in r.stdout, (r.stdout, r.stderr)
    m = re.search(r"scored (\d+) kmers (\d+) entries (\d+) eps ([0-9a-f]{8}) fnv ([0-9a-f]{16})", r.stdout)
    assert m, r.stdout
    mats = np.fromfile(dump, dtype=np.float32).reshape(6, 60, 4)
    eps = np.array([int(m.group(4), 16)], dtype=np.uint32).view(np.float32)[0]
    assert eps == np.float32(co.log_threshold(1.5, 4, 8))
    triples, scored = [], 0
    for g in range(3):
        keys, scores, emitted = co.explore_group(mats[2 * g:2 * g + 2], 8, float(eps))
        scored += emitted
        triples += [(int(k), 7 + g, int(s)) for k, s in zip(keys, scores.view(np.uint32))]
    triples.sort()
    fnv = 1469598103934665603
    for t in triples:
        for w in t:
            for i in range(4):
                fnv = ((fnv ^ ((w >> (8 * i)) & 0xFF)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert int(m.group(1)) == scored and int(m.group(3)) == len(triples) and int(m.group(2)) == len({t[0] for t in triples})
    assert int(m.group(5), 16) == fnv
