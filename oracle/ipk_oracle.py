"""ctypes front-end of oracle/libipk_oracle.so (the C restatement, oracle/ipk_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  PARITY UNPINNED (see the header of ipk_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libipk_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "ipk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libipk_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.ipko_bits.restype = C.c_uint
        L.ipko_bits.argtypes = [C.c_uint]
        L.ipko_prefix_max.restype = None
        L.ipko_prefix_max.argtypes = [f32p, C.c_size_t, C.c_uint, f32p]
        L.ipko_score_threshold.restype = C.c_float
        L.ipko_score_threshold.argtypes = [C.c_float, C.c_uint, C.c_uint]
        L.ipko_log_threshold.restype = C.c_float
        L.ipko_log_threshold.argtypes = [C.c_float, C.c_uint, C.c_uint]
        L.ipko_window.restype = C.c_size_t
        L.ipko_window.argtypes = [f32p, f32p, C.c_size_t, C.c_uint, C.c_uint, C.c_size_t, C.c_float,
                                  u32p, f32p, C.c_size_t]
        L.ipko_explore_group.restype = C.c_void_p
        L.ipko_explore_group.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_float]
        L.ipko_group_size.restype = C.c_size_t
        L.ipko_group_size.argtypes = [C.c_void_p]
        L.ipko_group_emitted.restype = C.c_uint64
        L.ipko_group_emitted.argtypes = [C.c_void_p]
        L.ipko_group_copy.restype = None
        L.ipko_group_copy.argtypes = [C.c_void_p, u32p, f32p]
        L.ipko_group_free.restype = None
        L.ipko_group_free.argtypes = [C.c_void_p]
        L.ipko_kmer_batch.restype = C.c_size_t
        L.ipko_kmer_batch.argtypes = [C.c_uint32, C.c_size_t]
        L.ipko_mif0.restype = C.c_double
        L.ipko_mif0.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_float]
        L.ipko_explore_many.restype = C.c_uint64
        L.ipko_explore_many.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint,
                                        C.c_float, C.POINTER(C.c_uint64)]
        L.ipko_explore_group_pos.restype = C.c_void_p
        L.ipko_explore_group_pos.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_float]
        L.ipko_group_pos_size.restype = C.c_size_t
        L.ipko_group_pos_size.argtypes = [C.c_void_p]
        L.ipko_group_pos_emitted.restype = C.c_uint64
        L.ipko_group_pos_emitted.argtypes = [C.c_void_p]
        L.ipko_group_pos_copy.restype = None
        L.ipko_group_pos_copy.argtypes = [C.c_void_p, u32p, f32p, u32p]
        L.ipko_group_pos_free.restype = None
        L.ipko_group_pos_free.argtypes = [C.c_void_p]
        L.ipko_log10f.restype = None
        L.ipko_log10f.argtypes = [f32p, C.c_size_t, f32p]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def bits(sigma):
    return int(lib().ipko_bits(sigma))


def log_threshold(omega, sigma, k):
    return float(lib().ipko_log_threshold(C.c_float(omega), sigma, k))


def score_threshold(omega, sigma, k):
    return float(lib().ipko_score_threshold(C.c_float(omega), sigma, k))


def prefix_max(m):
    """m: [sites, sigma] float32 -> best[sites+1] (matrix::preprocess)."""
    m, mp = _f32(m)
    sites, sigma = m.shape
    best = np.empty(sites + 1, dtype=np.float32)
    lib().ipko_prefix_max(mp, sites, sigma, best.ctypes.data_as(C.POINTER(C.c_float)))
    return best


def window(m, k, start, eps, best=None):
    """DCLA(window(m, start, k), k).run(eps): returns (keys u32, scores f32) sorted by key."""
    m, mp = _f32(m)
    sites, sigma = m.shape
    if best is None:
        best = prefix_max(m)
    best, bp = _f32(best)
    cap = 1024
    while True:
        keys = np.empty(cap, dtype=np.uint32)
        scores = np.empty(cap, dtype=np.float32)
        n = lib().ipko_window(mp, bp, sites, sigma, k, start, C.c_float(eps),
                              keys.ctypes.data_as(C.POINTER(C.c_uint32)),
                              scores.ctypes.data_as(C.POINTER(C.c_float)), cap)
        if n <= cap:
            break
        cap = n
    keys, scores = keys[:n], scores[:n]
    o = np.argsort(keys, kind="stable")
    return keys[o], scores[o]


def explore_group(mats, k, eps):
    """explore_group over mats [n_mats, sites, sigma]: (keys sorted, scores, emitted count)."""
    mats, mp = _f32(mats)
    n_mats, sites, sigma = mats.shape
    L = lib()
    g = L.ipko_explore_group(mp, n_mats, sites, sigma, k, C.c_float(eps))
    try:
        n = L.ipko_group_size(g)
        keys = np.empty(n, dtype=np.uint32)
        scores = np.empty(n, dtype=np.float32)
        L.ipko_group_copy(g, keys.ctypes.data_as(C.POINTER(C.c_uint32)),
                          scores.ctypes.data_as(C.POINTER(C.c_float)))
        emitted = int(L.ipko_group_emitted(g))
    finally:
        L.ipko_group_free(g)
    return keys, scores, emitted


def explore_many(mats, mats_per_group, k, eps):
    """Timing leg: mats [n_groups*mats_per_group, sites, sigma] -> (emitted, unique entries)."""
    mats, mp = _f32(mats)
    n, sites, sigma = mats.shape
    assert n % mats_per_group == 0
    uniq = C.c_uint64(0)
    e = lib().ipko_explore_many(mp, n // mats_per_group, mats_per_group, sites, sigma, k,
                                C.c_float(eps), C.byref(uniq))
    return int(e), int(uniq.value)


def kmer_batch(key, n):
    return int(lib().ipko_kmer_batch(key, n))


def mif0(log_scores, N, threshold):
    a, ap = _f32(log_scores)
    return float(lib().ipko_mif0(ap, a.size, N, C.c_float(threshold)))


def log10f(a):
    """libm log10f element-wise (numpy's own float32 log10 can differ in the last bit)."""
    a, ap = _f32(a)
    out = np.empty_like(a)
    lib().ipko_log10f(ap, a.size, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def explore_group_pos(mats, k, eps):
    """KEEP_POSITIONS flavour: (keys sorted, scores, positions, emitted)."""
    mats, mp = _f32(mats)
    n_mats, sites, sigma = mats.shape
    L = lib()
    g = L.ipko_explore_group_pos(mp, n_mats, sites, sigma, k, C.c_float(eps))
    try:
        n = L.ipko_group_pos_size(g)
        keys = np.empty(n, dtype=np.uint32); scores = np.empty(n, dtype=np.float32); pos = np.empty(n, dtype=np.uint32)
        L.ipko_group_pos_copy(g, keys.ctypes.data_as(C.POINTER(C.c_uint32)), scores.ctypes.data_as(C.POINTER(C.c_float)),
                              pos.ctypes.data_as(C.POINTER(C.c_uint32)))
        emitted = int(L.ipko_group_pos_emitted(g))
    finally:
        L.ipko_group_pos_free(g)
    return keys, scores, pos, emitted
