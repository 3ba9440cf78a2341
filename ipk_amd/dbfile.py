"""Database file: Python face of the C++ serialiser, plus a reader for the tests.

IPK streams its database through i2l::save_header / save_phylo_kmer over a Boost binary_oarchive
(ipk/src/db_builder.cpp:145-146,176-177,297-306,323-327).  The byte layout lives in ONE place,
ipk_amd/csrc/ipk_format.hpp (a reconstruction: i2l and Boost are un-vendored, so it is not pinned against a real .ipk);
it is written by ipkgpu_db_write (streamed from the device) and ipkgpu_db_write_host (host arrays: merged shards).
This module calls the latter and parses the layout back for the tests.
"""
import ctypes as C
import struct

import numpy as np

from .engine import IpkGpuError, load_library

ABI_SYMBOLS = ["ipkgpu_db_write", "ipkgpu_db_write_host", "ipkgpu_db_write_host_positions", "ipkgpu_db_write_last_error", "ipkgpu_db_write_time_s",
               "ipkgpu_db_merge_files", "ipkgpu_db_merge_last_error", "ipkgpu_db_protocol_version"]
_bound = False


class _Header(C.Structure):
    _fields_ = [("sequence_type", C.c_char_p), ("tree_index_size", C.c_uint64), ("tree_num_nodes", C.POINTER(C.c_uint32)),
                ("tree_subtree_length", C.POINTER(C.c_double)), ("newick", C.c_char_p), ("kmer_size", C.c_uint64),
                ("omega", C.c_float)]


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        L.ipkgpu_db_write.restype = C.c_int
        L.ipkgpu_db_write.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Header), C.c_char_p, C.POINTER(C.c_uint64)]
        L.ipkgpu_db_write_host.restype = C.c_int
        L.ipkgpu_db_write_host.argtypes = [C.POINTER(_Header), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_char_p, C.POINTER(C.c_uint64)]
        L.ipkgpu_db_write_host_positions.restype = C.c_int
        L.ipkgpu_db_write_host_positions.argtypes = [C.POINTER(_Header), C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        L.ipkgpu_db_write_last_error.restype = C.c_char_p
        L.ipkgpu_db_write_last_error.argtypes = []
        L.ipkgpu_db_write_time_s.restype = C.c_double
        L.ipkgpu_db_write_time_s.argtypes = [C.c_void_p, C.c_int]
        L.ipkgpu_db_merge_files.restype = C.c_int
        L.ipkgpu_db_merge_files.argtypes = [C.POINTER(_Header), C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ipkgpu_db_merge_last_error.restype = C.c_char_p
        L.ipkgpu_db_merge_last_error.argtypes = []
        L.ipkgpu_db_protocol_version.restype = C.c_uint32
        L.ipkgpu_db_protocol_version.argtypes = []
        _bound = True
    return L


def _header(sequence_type, tree_index, newick, kmer_size, omega):
    """tree_index: (num_nodes u32 [n], subtree_branch_length f64 [n]) or a list of (num_nodes, length) pairs."""
    if isinstance(tree_index, tuple) and len(tree_index) == 2 and hasattr(tree_index[0], "dtype"):
        nn, sl = tree_index
    else:
        nn = np.array([t[0] for t in tree_index], dtype=np.uint32)
        sl = np.array([t[1] for t in tree_index], dtype=np.float64)
    nn, sl = np.ascontiguousarray(nn, np.uint32), np.ascontiguousarray(sl, np.float64)
    h = _Header(sequence_type.encode(), len(nn), nn.ctypes.data_as(C.POINTER(C.c_uint32)), sl.ctypes.data_as(C.POINTER(C.c_double)),
                newick.encode(), int(kmer_size), float(omega))
    h._keep = (nn, sl)
    return h


def write_db_device(engine, db, path, sequence_type, tree_index, newick, kmer_size, omega):
    """ipkgpu_db_write: the shard `db` (filter values computed) streamed from device memory. Returns bytes written."""
    L = _lib()
    h = _header(sequence_type, tree_index, newick, kmer_size, omega)
    n = C.c_uint64()
    rc = L.ipkgpu_db_write(engine._h, db._h, C.byref(h), str(path).encode(), C.byref(n))
    if rc != 0:
        raise engine._err(rc)
    return int(n.value)


def write_times(engine):
    L = _lib()
    return {k: float(L.ipkgpu_db_write_time_s(engine._h, i)) for i, k in enumerate(("total_s", "device_s", "file_s"))}


def write_db(path, sequence_type, tree_index, newick, kmer_size, omega, keys, key_offsets, branches, scores, filter_values, order):
    """ipkgpu_db_write_host over host arrays: a database shard (ascending keys); order = positions in filter order."""
    L = _lib()
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    off = np.ascontiguousarray(key_offsets, dtype=np.uint64)
    ent = np.empty((len(branches), 2), dtype=np.uint32)
    ent[:, 0] = np.asarray(branches, dtype=np.uint32)
    ent[:, 1] = np.asarray(scores, dtype=np.float32).view(np.uint32)
    fv = np.ascontiguousarray(filter_values, dtype=np.float32)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    if len(off) == 0:
        off = np.zeros(1, np.uint64)
    h = _header(sequence_type, tree_index, newick, kmer_size, omega)
    n = C.c_uint64()
    rc = L.ipkgpu_db_write_host(C.byref(h), len(keys), keys.ctypes.data, off.ctypes.data, ent.ctypes.data, fv.ctypes.data,
                                order.ctypes.data, str(path).encode(), C.byref(n))
    if rc != 0:
        raise IpkGpuError(rc, L.ipkgpu_db_write_last_error().decode())
    return int(n.value)


def write_db_positions(path, sequence_type, tree_index, newick, kmer_size, omega, keys, key_offsets, branches, scores, positions,
                       filter_values, order):
    """ipkgpu_db_write_host_positions: the positioned database of ipk-aa-pos -- as write_db, every entry with the window position of
    its kept score (db_builder.cpp:655-662,687-689)."""
    L = _lib()
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    off = np.ascontiguousarray(key_offsets, dtype=np.uint64)
    ent = np.empty((len(branches), 2), dtype=np.uint32)
    ent[:, 0] = np.asarray(branches, dtype=np.uint32)
    ent[:, 1] = np.asarray(scores, dtype=np.float32).view(np.uint32)
    pos = np.ascontiguousarray(positions, dtype=np.uint32)
    fv = np.ascontiguousarray(filter_values, dtype=np.float32)
    order = np.ascontiguousarray(order, dtype=np.uint32)
    if len(off) == 0:
        off = np.zeros(1, np.uint64)
    h = _header(sequence_type, tree_index, newick, kmer_size, omega)
    n = C.c_uint64()
    rc = L.ipkgpu_db_write_host_positions(C.byref(h), len(keys), keys.ctypes.data, off.ctypes.data, ent.ctypes.data, pos.ctypes.data,
                                          fv.ctypes.data, order.ctypes.data, str(path).encode(), C.byref(n))
    if rc != 0:
        raise IpkGpuError(rc, L.ipkgpu_db_write_last_error().decode())
    return int(n.value)


def protocol_version():
    """What the writers put behind the archive preamble (0: no protocol word, no positions flag); ipk_format.hpp."""
    return int(_lib().ipkgpu_db_protocol_version())


def read_db(path, as_arrays=False, protocol=None):
    """Parses the layout of ipk_format.hpp.  Returns (header dict, records): records = list of (key, filter_value, branches,
    scores) in file order, or with as_arrays the arrays (keys, filter_values, counts, entry_offsets, branches, scores).
    protocol: the protocol version the file was written with (None: this process' -- IPKGPU_IPK_PROTOCOL_VERSION or the default)."""
    if protocol is None:
        protocol = protocol_version()
    raw = np.fromfile(path, dtype=np.uint8)
    buf = raw.tobytes()
    p = 0

    def take(fmt):
        nonlocal p
        v = struct.unpack_from("<" + fmt, buf, p)
        p += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def take_string():
        nonlocal p
        n = take("Q")
        s = buf[p:p + n].decode()
        p += n
        return s

    assert take_string() == "serialization::archive", "not a Boost binary archive preamble"
    lib_version = take("H")
    sizes = take("BBBB")
    assert sizes == (4, 8, 4, 8) and take("i") == 1
    proto = take("I") if protocol else 0
    assert proto == protocol, f"protocol word {proto}, expected {protocol}"
    st = take_string()
    positions = bool(take("B")) if protocol else False
    ni = take("Q")
    ti = np.frombuffer(buf, dtype=[("n", "<u8"), ("l", "<f8")], count=ni, offset=p)
    p += ni * 16
    newick = take_string()
    k = take("Q"); omega = take("f"); nk = take("Q"); ne = take("Q")
    hdr = dict(sequence_type=st, tree_index=[(int(a), float(b)) for a, b in ti], newick=newick, kmer_size=k, omega=omega,
               total_num_kmers=nk, total_num_entries=ne, library_version=lib_version, protocol_version=proto,
               positions_loaded=positions)
    if positions:
        # records of 16 + 10 n bytes: key, filter value, n, then n x (u32 branch, f32 score, u16 position)
        recs, q = [], p
        ent_t = np.dtype([("b", "<u4"), ("s", "<f4"), ("p", "<u2")])
        for _ in range(nk):
            key, fvb, cnt = struct.unpack_from("<IfQ", buf, q)
            e = np.frombuffer(buf, dtype=ent_t, count=cnt, offset=q + 16)
            recs.append((key, fvb, e["b"].copy(), e["s"].copy(), e["p"].astype(np.uint32)))
            q += 16 + 10 * cnt
        assert q == len(buf) and sum(len(r[2]) for r in recs) == ne, "body size does not match the header's totals"
        return hdr, recs
    body = np.frombuffer(buf[p:], dtype="<u4") if (len(buf) - p) % 4 == 0 else None
    assert body is not None and len(body) == 4 * nk + 2 * ne, "body size does not match the header's totals"
    # record starts: head of 4 words, then 2 words per entry -- walk the counts
    keys = np.empty(nk, np.uint32); fvs = np.empty(nk, np.float32); counts = np.empty(nk, np.uint64)
    starts = np.empty(nk, np.int64)
    q = 0
    for i in range(nk):
        starts[i] = q
        c = int(body[q + 2]) | (int(body[q + 3]) << 32)
        counts[i] = c
        q += 4 + 2 * c
    assert q == len(body)
    keys[:] = body[starts]; fvs[:] = body[starts + 1].view(np.float32)
    eoff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    mask = np.ones(len(body), bool)
    for d in range(4):
        mask[starts + d] = False
    ent = body[mask].reshape(-1, 2)
    br, sc = ent[:, 0].copy(), ent[:, 1].copy().view(np.float32)
    if as_arrays:
        return hdr, (keys, fvs, counts, eoff, br, sc)
    recs = [(int(keys[i]), float(fvs[i]), br[eoff[i]:eoff[i + 1]], sc[eoff[i]:eoff[i + 1]]) for i in range(nk)]
    return hdr, recs


# ---- several GPUs: every rank owns the k-mers with code % P == rank ---------------------------------
# Filter values are per k-mer, so each rank computes them on its own shard and writes the shard, in its own filter order, as a
# database file; the final file is the streaming merge of the P shard files by filter value (ipkgpu_db_merge_files) -- the
# role merge_stage2 plays for the reference's on-disk batches (db_builder.cpp:392-458).

def filter_sort_code(filter_values, keys):
    """The engine's filter order as one integer per k-mer: order-preserving code of the float32 filter value, ties by
    ascending key (ipk_amd/csrc/kernels_filter.hpp, filter_sortkey_kernel)."""
    u = np.asarray(filter_values, dtype=np.float32).view(np.uint32).astype(np.uint64)
    code = np.where(u & np.uint64(0x80000000), ~u & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (code << np.uint64(32)) | np.asarray(keys, dtype=np.uint64)


def splitmix_unit(keys):
    """A reproducible value in [0, 1) per k-mer code (the `random` filter): independent of the sharding."""
    x = np.asarray(keys, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def merge_shard_files(path, sequence_type, tree_index, newick, kmer_size, omega, shard_paths):
    """ipkgpu_db_merge_files: streaming P-way merge of the ranks' shard files (each a database file in its own filter order)
    by (filter value, key) -- merge_stage2's role (db_builder.cpp:392-458).  Returns (total k-mers, total entries)."""
    L = _lib()
    h = _header(sequence_type, tree_index, newick, kmer_size, omega)
    arr = (C.c_char_p * len(shard_paths))(*[str(p).encode() for p in shard_paths])
    nk, ne, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.ipkgpu_db_merge_files(C.byref(h), arr, len(shard_paths), str(path).encode(), C.byref(nk), C.byref(ne), C.byref(nb))
    if rc != 0:
        raise IpkGpuError(rc, L.ipkgpu_db_merge_last_error().decode())
    return int(nk.value), int(ne.value)
