"""CPU suite: database container round trip (row n2; format parity with i2l's .ipk is unpinned)."""
import os

import numpy as np
import pytest

import ipk_amd

from ipk_amd import dbfile
from ipk_amd.synth import synth_matrices
from oracle import db_oracle as dbo
from oracle import ipk_oracle as co


@pytest.mark.parametrize("protocol", [None, "0", "5"])
def test_round_trip(tmp_path, monkeypatch, protocol):
    """protocol: the guessed protocol-version word + positions flag (ipk_format.hpp; a loaded database answers version() and
    positions_loaded(), tools/src/diff.cpp:41-46,137-145) -- default on, IPKGPU_IPK_PROTOCOL_VERSION=0 leaves both out."""
    if protocol is not None:
        monkeypatch.setenv("IPKGPU_IPK_PROTOCOL_VERSION", protocol)
    want = 7 if protocol is None else int(protocol)
    assert dbfile.protocol_version() == want
    sigma, k = 4, 6
    mats = synth_matrices(6, 30, sigma, 0.2, 3)
    eps = co.log_threshold(1.5, sigma, k)
    thr = co.score_threshold(1.5, sigma, k)
    full = dbo.build_db([(10 + g,) + co.explore_group(mats[2 * g:2 * g + 2], k, eps)[:2] for g in range(3)])
    keys, off, br, sc = dbo.db_shard_arrays(full, sigma, k, 0, 1)
    fv = np.array([co.mif0(sc[int(off[i]):int(off[i + 1])].view(np.float32), 5, thr) for i in range(len(keys))], dtype=np.float32)
    order = np.lexsort((keys, fv))                    # filter value, ties by key (db_builder.cpp:284)
    path = tmp_path / "db.ipkgpu"
    dbfile.write_db(path, "DNA", [(5, 0.5), (3, 0.25), (1, 0.0)], "((A:1,B:1):1,C:2);", k, 1.5, keys, off, br,
                    sc.view(np.float32), fv, order)
    hdr, recs = dbfile.read_db(path)
    assert hdr["protocol_version"] == want and hdr["positions_loaded"] is False
    head_bytes = os.path.getsize(path) - 16 * len(keys) - 8 * len(br)
    assert head_bytes == 40 + (5 if want else 0) + 8 + 3 + 8 + 3 * 16 + 8 + len("((A:1,B:1):1,C:2);") + 8 + 4 + 8 + 8
    assert hdr["sequence_type"] == "DNA" and hdr["kmer_size"] == k and hdr["total_num_kmers"] == len(keys)
    assert hdr["total_num_entries"] == len(br) and hdr["newick"].startswith("((A") and hdr["tree_index"][0] == (5, 0.5)
    assert [r[0] for r in recs] == keys[order].tolist()
    assert all(recs[i][1] <= recs[i + 1][1] for i in range(len(recs) - 1))
    for r in recs[:50]:
        assert [(int(b), int(s)) for b, s in zip(r[2], r[3].view(np.uint32))] == full[r[0]]


def test_positioned_round_trip(tmp_path, monkeypatch):
    """ipk-aa-pos (KEEP_POSITIONS, db_builder.cpp:655-662,687-689): entries of (branch, score, position), the header's positions flag
    set; a position that does not fit the field, or a layout without the flag, is refused."""
    rng = np.random.default_rng(11)
    n = 200
    keys = np.sort(rng.choice(20 ** 4, size=n, replace=False)).astype(np.uint32)
    lens = rng.integers(1, 5, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    br = rng.integers(0, 30, size=int(off[-1])).astype(np.uint32)
    sc = (-rng.random(int(off[-1])) * 4).astype(np.float32)
    pos = rng.integers(0, 3000, size=int(off[-1])).astype(np.uint32)
    fv = rng.normal(size=n).astype(np.float32)
    order = np.argsort(dbfile.filter_sort_code(fv, keys), kind="stable")
    path = tmp_path / "pos.ipk"
    nbytes = dbfile.write_db_positions(path, "AA", [(3, 0.5)], "(a,b);", 4, 1.5, keys, off, br, sc, pos, fv, order)
    assert nbytes == os.path.getsize(path)
    hdr, recs = dbfile.read_db(path)
    assert hdr["positions_loaded"] is True and hdr["sequence_type"] == "AA" and hdr["total_num_entries"] == len(br)
    assert [r[0] for r in recs] == keys[order].tolist()
    for r, i in zip(recs, order.tolist()):
        a, b = int(off[i]), int(off[i + 1])
        assert np.array_equal(r[2], br[a:b]) and np.array_equal(r[3].view(np.uint32), sc[a:b].view(np.uint32)) and np.array_equal(r[4], pos[a:b])
    with pytest.raises(ipk_amd.IpkGpuError):
        dbfile.write_db_positions(tmp_path / "x.ipk", "AA", [], "", 4, 1.5, keys, off, br, sc, pos + 70000, fv, order)
    monkeypatch.setenv("IPKGPU_IPK_PROTOCOL_VERSION", "0")
    with pytest.raises(ipk_amd.IpkGpuError):
        dbfile.write_db_positions(tmp_path / "y.ipk", "AA", [], "", 4, 1.5, keys, off, br, sc, pos, fv, order)


def test_merge_of_rank_shards_equals_the_single_writer(tmp_path, monkeypatch):
    """Several GPUs: rank r owns the k-mers with code % P == r and its own filter values; merging the shard files by
    (filter value, key) must give the file one GPU writes (ipkgpu_db_merge_files: the role of merge_stage2,
    db_builder.cpp:392-458).  A shard is a database file in the rank's own filter order; an empty shard takes part too."""
    rng = np.random.default_rng(5)
    n = 500
    keys = np.sort(rng.choice(4 ** 8, size=n, replace=False)).astype(np.uint32)
    lens = rng.integers(1, 6, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    br = rng.integers(0, 40, size=int(off[-1])).astype(np.uint32)
    sc = (-rng.random(int(off[-1])) * 5).astype(np.float32)
    fv = np.round(rng.normal(size=n), 1).astype(np.float32)          # coarse values: plenty of ties, both signs
    fv[::50] = 0.0; fv[25::50] = -0.0                                 # and both zeros
    order = np.argsort(dbfile.filter_sort_code(fv, keys), kind="stable")
    one = tmp_path / "one.db"
    dbfile.write_db(one, "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, keys, off, br, sc, fv, order)
    with monkeypatch.context() as m:                                  # a shard written under another protocol word is not this layout
        m.setenv("IPKGPU_IPK_PROTOCOL_VERSION", "0")
        with pytest.raises(ipk_amd.IpkGpuError):
            dbfile.merge_shard_files(tmp_path / "x.db", "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, [one])
    for world in (2, 3):
        paths = []
        for r in range(world):
            m = keys % world == r
            ml = lens[m]
            idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in np.nonzero(m)[0]]).astype(np.int64)
            so = np.concatenate([[0], np.cumsum(ml)]).astype(np.uint64)
            paths.append(tmp_path / f"s{world}_{r}.ipk")
            so_r = np.argsort(dbfile.filter_sort_code(fv[m], keys[m]), kind="stable")
            dbfile.write_db(paths[-1], "DNA", [], "", 8, 1.5, keys[m], so, br[idx], sc[idx], fv[m], so_r)
        paths.append(tmp_path / f"s{world}_empty.ipk")                # a rank that owns no k-mer
        dbfile.write_db(paths[-1], "DNA", [], "", 8, 1.5, keys[:0], np.zeros(1, np.uint64), br[:0], sc[:0], fv[:0], np.zeros(0, np.uint32))
        merged = tmp_path / f"merged{world}.db"
        assert dbfile.merge_shard_files(merged, "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, paths) == (n, int(off[-1]))
        assert open(merged, "rb").read() == open(one, "rb").read()
    with pytest.raises(ipk_amd.IpkGpuError):                          # a truncated shard is an error, not a short file
        bad = tmp_path / "bad.ipk"
        bad.write_bytes(open(paths[0], "rb").read()[:-5])
        dbfile.merge_shard_files(tmp_path / "x.db", "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, [bad, paths[1]])


_MERGE_RSS_SCRIPT = r"""
import ctypes as C, resource, sys
lib = C.CDLL(sys.argv[1])
class H(C.Structure):
    _fields_ = [("sequence_type", C.c_char_p), ("tree_index_size", C.c_uint64), ("tree_num_nodes", C.c_void_p),
                ("tree_subtree_length", C.c_void_p), ("newick", C.c_char_p), ("kmer_size", C.c_uint64), ("omega", C.c_float)]
h = H(b"DNA", 0, None, None, b"(a,b);", 10, 1.5)
paths = [p.encode() for p in sys.argv[3:]]
arr = (C.c_char_p * len(paths))(*paths)
nk, ne, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
lib.ipkgpu_db_merge_files.argtypes = [C.POINTER(H), C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
rc = lib.ipkgpu_db_merge_files(C.byref(h), arr, len(paths), sys.argv[2].encode(), C.byref(nk), C.byref(ne), C.byref(nb))
after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("MERGE", rc, nk.value, ne.value, nb.value, before, after)
"""


def test_merge_streams_with_bounded_memory(tmp_path):
    """Two shards of ~100 MB each: the merge's resident set grows by its buffers (a few MiB per shard + the output buffer),
    not by the shards' size -- rank 0 of an 8-GPU cfg3 build would otherwise hold tens of GB of entries."""
    import subprocess
    import sys
    rng = np.random.default_rng(11)
    n = 1_500_000
    paths, tot_e = [], 0
    for r in range(2):
        keys = (np.arange(n, dtype=np.uint32) * 2 + r)
        lens = rng.integers(4, 12, size=n)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        ne = int(off[-1]); tot_e += ne
        fv = rng.normal(size=n).astype(np.float32)
        order = np.argsort(dbfile.filter_sort_code(fv, keys), kind="stable").astype(np.uint32)
        paths.append(str(tmp_path / f"big{r}.ipk"))
        dbfile.write_db(paths[-1], "DNA", [], "", 10, 1.5, keys, off, np.zeros(ne, np.uint32), np.zeros(ne, np.float32), fv, order)
        assert os.path.getsize(paths[-1]) > 90 << 20
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ipk_amd", "libipkgpu.so")
    out = subprocess.run([sys.executable, "-c", _MERGE_RSS_SCRIPT, lib, str(tmp_path / "merged.ipk")] + paths,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    f = [l for l in out.stdout.splitlines() if l.startswith("MERGE")][0].split()
    rc, nk, ne, nb, before, after = (int(x) for x in f[1:])
    assert rc == 0 and nk == 2 * n and ne == tot_e
    assert nb == os.path.getsize(tmp_path / "merged.ipk") > 180 << 20
    assert (after - before) < 64 * 1024, f"resident set grew by {(after - before) / 1024:.0f} MiB while merging {nb >> 20} MiB"   # ru_maxrss is in KiB
    # and the records came out in (filter value, key) order
    hdr, (keys, fvs, counts, eoff, br, sc) = dbfile.read_db(tmp_path / "merged.ipk", as_arrays=True)
    code = dbfile.filter_sort_code(fvs, keys)
    assert np.all(code[1:] > code[:-1])
