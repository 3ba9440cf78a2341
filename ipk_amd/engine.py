"""ctypes binding of libipkgpu.so (include/ipkgpu.h).

Mirrors the seam of ipk/src/db_builder.cpp:576-698: ``Engine.score_groups`` takes the ghost-node
matrices of a batch of branch groups and returns, per branch id, the max-reduced (key, score) set
that ``explore_group`` leaves in ``group_map`` (db_builder.cpp:685), plus the number of scored
phylo-k-mers (the reference's ``count``, db_builder.cpp:664).
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libipkgpu.so")
_lib = None

T_TOTAL, T_PREFIX, T_SCORE, T_COMPACT, T_SCORE_LAUNCHES, T_SCORE_MAIN, T_SCORE_REDUCE, T_XP_COUNT, T_XP_WRITE, T_KM_WRITE = range(10)

# every symbol include/ipkgpu.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ipkgpu_create", "ipkgpu_destroy", "ipkgpu_last_error", "ipkgpu_last_main_kernel", "ipkgpu_last_tables_compressed", "ipkgpu_set_option",
    "ipkgpu_log_threshold", "ipkgpu_bits_per_symbol", "ipkgpu_kmer_batch", "ipkgpu_max_k",
    "ipkgpu_score_groups", "ipkgpu_score_groups_device",
    "ipkgpu_result_num_groups", "ipkgpu_result_group_ids", "ipkgpu_result_offsets",
    "ipkgpu_result_emitted", "ipkgpu_result_keys", "ipkgpu_result_scores",
    "ipkgpu_result_keys_device", "ipkgpu_result_scores_device", "ipkgpu_result_time_ms",
    "ipkgpu_result_free", "ipkgpu_score_groups_positions", "ipkgpu_result_positions", "ipkgpu_debug_exec_violations",
]


class IpkGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ipkgpu error {code}: {message}")
        self.code = code


def load_library():
    """Loads libipkgpu.so (fails loudly if the HIP extension has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: build it with `python -m ipk_amd.build` "
                          "(there is no CPU fallback)")
    # torch ships its own HIP runtime under the same SONAME; import it first so that one copy
    # of libamdhip64 serves both torch tensors and this library.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    # IPKGPU_LIB: an alternative build of the same library (tuning experiments only)
    L = C.CDLL(os.environ.get("IPKGPU_LIB") or _LIB_PATH)
    f32p, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.ipkgpu_create.restype = C.c_int
    L.ipkgpu_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.ipkgpu_destroy.restype = None
    L.ipkgpu_destroy.argtypes = [C.c_void_p]
    L.ipkgpu_last_error.restype = C.c_char_p
    L.ipkgpu_last_error.argtypes = [C.c_void_p]
    L.ipkgpu_last_main_kernel.restype = C.c_char_p
    L.ipkgpu_last_main_kernel.argtypes = [C.c_void_p]
    L.ipkgpu_last_tables_compressed.restype = C.c_int
    L.ipkgpu_last_tables_compressed.argtypes = [C.c_void_p]
    L.ipkgpu_debug_exec_violations.restype = C.c_int64
    L.ipkgpu_debug_exec_violations.argtypes = [C.c_void_p]
    L.ipkgpu_set_option.restype = C.c_int
    L.ipkgpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ipkgpu_log_threshold.restype = C.c_float
    L.ipkgpu_log_threshold.argtypes = [C.c_float, C.c_uint32, C.c_uint32]
    L.ipkgpu_bits_per_symbol.restype = C.c_uint32
    L.ipkgpu_bits_per_symbol.argtypes = [C.c_uint32]
    L.ipkgpu_kmer_batch.restype = C.c_size_t
    L.ipkgpu_kmer_batch.argtypes = [C.c_uint32, C.c_size_t]
    L.ipkgpu_max_k.restype = C.c_uint32
    L.ipkgpu_max_k.argtypes = [C.c_uint32]
    for name in ("ipkgpu_score_groups", "ipkgpu_score_groups_device"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32,
                       C.c_float, C.POINTER(C.c_void_p)]
    L.ipkgpu_result_num_groups.restype = C.c_uint32
    L.ipkgpu_result_num_groups.argtypes = [C.c_void_p]
    L.ipkgpu_result_group_ids.restype = u32p
    L.ipkgpu_result_group_ids.argtypes = [C.c_void_p]
    L.ipkgpu_result_offsets.restype = u64p
    L.ipkgpu_result_offsets.argtypes = [C.c_void_p]
    L.ipkgpu_result_emitted.restype = C.c_uint64
    L.ipkgpu_result_emitted.argtypes = [C.c_void_p]
    L.ipkgpu_result_keys.restype = u32p
    L.ipkgpu_result_keys.argtypes = [C.c_void_p]
    L.ipkgpu_result_scores.restype = f32p
    L.ipkgpu_result_scores.argtypes = [C.c_void_p]
    L.ipkgpu_result_keys_device.restype = C.c_void_p
    L.ipkgpu_result_keys_device.argtypes = [C.c_void_p]
    L.ipkgpu_result_scores_device.restype = C.c_void_p
    L.ipkgpu_result_scores_device.argtypes = [C.c_void_p]
    L.ipkgpu_result_time_ms.restype = C.c_double
    L.ipkgpu_result_time_ms.argtypes = [C.c_void_p, C.c_int]
    L.ipkgpu_result_free.restype = None
    L.ipkgpu_result_free.argtypes = [C.c_void_p]
    L.ipkgpu_score_groups_positions.restype = C.c_int
    L.ipkgpu_score_groups_positions.argtypes = L.ipkgpu_score_groups.argtypes
    L.ipkgpu_result_positions.restype = u32p
    L.ipkgpu_result_positions.argtypes = [C.c_void_p]
    _lib = L
    return L


def log_threshold(omega, sigma, k):
    return float(load_library().ipkgpu_log_threshold(C.c_float(omega), sigma, k))


def bits_per_symbol(sigma):
    return int(load_library().ipkgpu_bits_per_symbol(sigma))


def score_threshold(omega, sigma, k):
    L = load_library()
    _bind_keymajor(L)
    return float(L.ipkgpu_score_threshold(C.c_float(omega), sigma, k))


def kmer_batch(key, n_ranges):
    return int(load_library().ipkgpu_kmer_batch(key, n_ranges))


def max_k(sigma):
    return int(load_library().ipkgpu_max_k(sigma))


class Result:
    """Owner of an ipkgpu_result (CSR of per-branch (key, score) sets)."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle
        n = lib.ipkgpu_result_num_groups(handle)
        self.group_ids = np.ctypeslib.as_array(lib.ipkgpu_result_group_ids(handle), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        self.offsets = np.ctypeslib.as_array(lib.ipkgpu_result_offsets(handle), shape=(n + 1,)).copy()
        self.emitted = int(lib.ipkgpu_result_emitted(handle))

    @property
    def num_entries(self):
        return int(self.offsets[-1])

    def time_ms(self, which):
        return float(self._lib.ipkgpu_result_time_ms(self._h, which))

    def keys(self):
        n = self.num_entries
        if n == 0:
            return np.zeros(0, np.uint32)
        p = self._lib.ipkgpu_result_keys(self._h)
        if not p:
            raise IpkGpuError(2, "device-to-host copy of keys failed")
        return np.ctypeslib.as_array(p, shape=(n,))

    def scores(self):
        n = self.num_entries
        if n == 0:
            return np.zeros(0, np.float32)
        p = self._lib.ipkgpu_result_scores(self._h)
        if not p:
            raise IpkGpuError(2, "device-to-host copy of scores failed")
        return np.ctypeslib.as_array(p, shape=(n,))

    def positions(self):
        """Window positions aligned with keys()/scores() (KEEP_POSITIONS results only)."""
        n = self.num_entries
        if n == 0:
            return np.zeros(0, np.uint32)
        p = self._lib.ipkgpu_result_positions(self._h)
        if not p:
            raise IpkGpuError(1, "this result carries no positions")
        return np.ctypeslib.as_array(p, shape=(n,))

    def keys_device_ptr(self):
        return self._lib.ipkgpu_result_keys_device(self._h)

    def scores_device_ptr(self):
        return self._lib.ipkgpu_result_scores_device(self._h)

    def group(self, i):
        """(keys, scores) of the i-th group in first-seen order (host copies)."""
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        return self.keys()[a:b], self.scores()[a:b]

    def free(self):
        if self._h:
            self._lib.ipkgpu_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One scoring context on one GPU (one process per GPU)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.ipkgpu_create(device_id, C.byref(h))
        if rc != 0:
            raise IpkGpuError(rc, self._lib.ipkgpu_last_error(None).decode())
        self._h = h

    def _err(self, rc):
        return IpkGpuError(rc, self._lib.ipkgpu_last_error(self._h).decode())

    def last_tables_compressed(self):
        return bool(self._lib.ipkgpu_last_tables_compressed(self._h))

    def last_main_kernel(self):
        return self._lib.ipkgpu_last_main_kernel(self._h).decode()

    def set_option(self, name, value):
        rc = self._lib.ipkgpu_set_option(self._h, name.encode(), int(value))
        if rc != 0:
            raise self._err(rc)

    def score_groups(self, logp, mat_group, k, log_eps, sigma=None, sites=None, n_mats=None):
        """logp: numpy float32 [n_mats, sites, sigma] (host path) or a torch CUDA tensor / raw
        device pointer with explicit n_mats/sites/sigma (device path)."""
        mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
        out = C.c_void_p()
        gp = mat_group.ctypes.data_as(C.POINTER(C.c_uint32))
        if isinstance(logp, np.ndarray):
            logp = np.ascontiguousarray(logp, dtype=np.float32)
            n_mats, sites, sigma = logp.shape
            if mat_group.shape != (n_mats,):
                raise ValueError("mat_group must have one branch id per matrix")
            rc = self._lib.ipkgpu_score_groups(self._h, logp.ctypes.data_as(C.c_void_p), n_mats, sites, sigma,
                                               gp, k, C.c_float(log_eps), C.byref(out))
        else:
            if hasattr(logp, "data_ptr"):
                if not logp.is_cuda or not logp.is_contiguous() or logp.dtype.itemsize != 4:
                    raise ValueError("device path needs a contiguous float32 CUDA tensor")
                n_mats, sites, sigma = logp.shape
                ptr = logp.data_ptr()
                # the library works on its own stream: whatever produced `logp` on torch's stream must be complete
                import torch
                torch.cuda.current_stream().synchronize()
            else:
                ptr = int(logp)
            if mat_group.shape != (n_mats,):
                raise ValueError("mat_group must have one branch id per matrix")
            rc = self._lib.ipkgpu_score_groups_device(self._h, C.c_void_p(ptr), n_mats, sites, sigma,
                                                      gp, k, C.c_float(log_eps), C.byref(out))
        if rc != 0:
            raise self._err(rc)
        return self._adopt(Result(self._lib, out))

    def score_groups_positions(self, logp, mat_group, k, log_eps):
        """KEEP_POSITIONS flavour (ipk-aa-pos): logp numpy float32 [n_mats, sites, sigma]; Result.positions()."""
        logp = np.ascontiguousarray(logp, dtype=np.float32)
        mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
        n_mats, sites, sigma = logp.shape
        if mat_group.shape != (n_mats,):
            raise ValueError("mat_group must have one branch id per matrix")
        out = C.c_void_p()
        rc = self._lib.ipkgpu_score_groups_positions(self._h, logp.ctypes.data_as(C.c_void_p), n_mats, sites, sigma,
                                                     mat_group.ctypes.data_as(C.POINTER(C.c_uint32)), k, C.c_float(log_eps),
                                                     C.byref(out))
        if rc != 0:
            raise self._err(rc)
        return self._adopt(Result(self._lib, out))

    def _adopt(self, obj):
        """Results, parts and databases hold device blocks of this context: close() frees the ones still alive first (a handle freed
        after its context is a crash, and a failing test leaves exactly that order to the garbage collector)."""
        kids = self.__dict__.setdefault("_children", weakref.WeakSet())
        kids.add(obj)
        return obj

    def close(self):
        if self._h:
            for kid in list(self.__dict__.get("_children", ())):
                try:
                    kid.free()
                except Exception:
                    pass
            self._lib.ipkgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- key-major parts / database shards (multi-GPU exchange step) ---------------------------------

def _bind_keymajor(L):
    if getattr(L, "_km_bound", False):
        return
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.ipkgpu_score_groups_keymajor_device.restype = C.c_int
    L.ipkgpu_score_groups_keymajor_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u32p,
                                                      C.c_uint32, C.c_float, C.c_uint32, C.POINTER(C.c_void_p)]
    L.ipkgpu_parts_num_owners.restype = C.c_uint32
    L.ipkgpu_parts_num_owners.argtypes = [C.c_void_p]
    L.ipkgpu_parts_slots.restype = C.c_uint64
    L.ipkgpu_parts_slots.argtypes = [C.c_void_p]
    L.ipkgpu_parts_counts_device.restype = C.c_void_p
    L.ipkgpu_parts_counts_device.argtypes = [C.c_void_p]
    L.ipkgpu_parts_entries_device.restype = C.c_void_p
    L.ipkgpu_parts_entries_device.argtypes = [C.c_void_p]
    L.ipkgpu_parts_owner_offsets.restype = u64p
    L.ipkgpu_parts_owner_offsets.argtypes = [C.c_void_p]
    L.ipkgpu_parts_emitted.restype = C.c_uint64
    L.ipkgpu_parts_emitted.argtypes = [C.c_void_p]
    L.ipkgpu_parts_time_ms.restype = C.c_double
    L.ipkgpu_parts_time_ms.argtypes = [C.c_void_p, C.c_int]
    L.ipkgpu_parts_free.restype = None
    L.ipkgpu_parts_free.argtypes = [C.c_void_p]
    L.ipkgpu_merge_parts.restype = C.c_int
    L.ipkgpu_merge_parts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p, u64p, C.POINTER(C.c_void_p)]
    L.ipkgpu_merge_parts_ptrs.restype = C.c_int
    L.ipkgpu_merge_parts_ptrs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.ipkgpu_comm_unique_id.restype = C.c_int
    L.ipkgpu_comm_unique_id.argtypes = [C.c_void_p]
    L.ipkgpu_comm_init.restype = C.c_int
    L.ipkgpu_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ipkgpu_comm_available.restype = C.c_int
    L.ipkgpu_comm_available.argtypes = []
    L.ipkgpu_comm_prepare.restype = C.c_int
    L.ipkgpu_comm_prepare.argtypes = [C.c_void_p, C.c_int]
    for n in ("ipkgpu_comm_rank", "ipkgpu_comm_world"):
        getattr(L, n).restype = C.c_int
        getattr(L, n).argtypes = [C.c_void_p]
    L.ipkgpu_exchange_begin.restype = C.c_int
    L.ipkgpu_exchange_begin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.ipkgpu_exchange_merge.restype = C.c_int
    L.ipkgpu_exchange_merge.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_double)]
    L.ipkgpu_xfer_exposed_ms.restype = C.c_double
    L.ipkgpu_xfer_exposed_ms.argtypes = [C.c_void_p]
    L.ipkgpu_xfer_free.restype = None
    L.ipkgpu_xfer_free.argtypes = [C.c_void_p]
    L.ipkgpu_db_from_parts.restype = C.c_int
    L.ipkgpu_db_from_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.ipkgpu_db_num_keys.restype = C.c_uint64
    L.ipkgpu_db_num_keys.argtypes = [C.c_void_p]
    L.ipkgpu_db_num_entries.restype = C.c_uint64
    L.ipkgpu_db_num_entries.argtypes = [C.c_void_p]
    L.ipkgpu_db_keys.restype = u32p
    L.ipkgpu_db_keys.argtypes = [C.c_void_p]
    L.ipkgpu_db_key_offsets.restype = u64p
    L.ipkgpu_db_key_offsets.argtypes = [C.c_void_p]
    L.ipkgpu_db_entries.restype = u32p
    L.ipkgpu_db_entries.argtypes = [C.c_void_p]
    for n in ("ipkgpu_db_keys_device", "ipkgpu_db_key_offsets_device", "ipkgpu_db_entries_device"):
        getattr(L, n).restype = C.c_void_p
        getattr(L, n).argtypes = [C.c_void_p]
    L.ipkgpu_db_time_ms.restype = C.c_double
    L.ipkgpu_db_time_ms.argtypes = [C.c_void_p]
    L.ipkgpu_db_free.restype = None
    L.ipkgpu_db_free.argtypes = [C.c_void_p]
    L.ipkgpu_score_threshold.restype = C.c_float
    L.ipkgpu_score_threshold.argtypes = [C.c_float, C.c_uint32, C.c_uint32]
    L.ipkgpu_db_filter_mif0.restype = C.c_int
    L.ipkgpu_db_filter_mif0.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float]
    L.ipkgpu_db_filter_values.restype = C.POINTER(C.c_float)
    L.ipkgpu_db_filter_values.argtypes = [C.c_void_p]
    L.ipkgpu_db_filter_values_f64.restype = C.POINTER(C.c_double)
    L.ipkgpu_db_filter_values_f64.argtypes = [C.c_void_p]
    L.ipkgpu_db_filter_order.restype = u32p
    L.ipkgpu_db_filter_order.argtypes = [C.c_void_p]
    for n in ("ipkgpu_db_filter_values_device", "ipkgpu_db_filter_order_device"):
        getattr(L, n).restype = C.c_void_p
        getattr(L, n).argtypes = [C.c_void_p]
    L.ipkgpu_db_filter_time_ms.restype = C.c_double
    L.ipkgpu_db_filter_time_ms.argtypes = [C.c_void_p]
    L._km_bound = True


ABI_SYMBOLS += [
    "ipkgpu_score_groups_keymajor_device", "ipkgpu_parts_num_owners", "ipkgpu_parts_slots",
    "ipkgpu_parts_counts_device", "ipkgpu_parts_entries_device", "ipkgpu_parts_owner_offsets",
    "ipkgpu_parts_emitted", "ipkgpu_parts_time_ms", "ipkgpu_parts_free", "ipkgpu_merge_parts",
    "ipkgpu_db_num_keys", "ipkgpu_db_num_entries", "ipkgpu_db_keys", "ipkgpu_db_key_offsets", "ipkgpu_db_entries",
    "ipkgpu_db_keys_device", "ipkgpu_db_key_offsets_device", "ipkgpu_db_entries_device", "ipkgpu_db_time_ms",
    "ipkgpu_db_free", "ipkgpu_db_from_parts",
    "ipkgpu_score_threshold", "ipkgpu_db_filter_mif0", "ipkgpu_db_filter_values", "ipkgpu_db_filter_values_f64",
    "ipkgpu_db_filter_order", "ipkgpu_db_filter_values_device", "ipkgpu_db_filter_order_device",
    "ipkgpu_db_filter_time_ms",
    "ipkgpu_merge_parts_ptrs", "ipkgpu_comm_unique_id", "ipkgpu_comm_init", "ipkgpu_comm_rank", "ipkgpu_comm_world",
    "ipkgpu_exchange_begin", "ipkgpu_exchange_merge", "ipkgpu_xfer_exposed_ms", "ipkgpu_xfer_free",
    "ipkgpu_comm_available", "ipkgpu_comm_prepare",
]


class Parts:
    """This rank's key-major partial database, split by owner (device resident)."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle
        self.n_owners = int(lib.ipkgpu_parts_num_owners(handle))
        self.slots = int(lib.ipkgpu_parts_slots(handle))
        self.owner_offsets = np.ctypeslib.as_array(lib.ipkgpu_parts_owner_offsets(handle), shape=(self.n_owners + 1,)).copy()
        self.emitted = int(lib.ipkgpu_parts_emitted(handle))
        self.extra_ms = {}        # timings of further pieces folded into this object (distributed.build_db_shard)

    @property
    def num_entries(self):
        return int(self.owner_offsets[-1])

    def counts_ptr(self):
        return self._lib.ipkgpu_parts_counts_device(self._h)

    def entries_ptr(self):
        return self._lib.ipkgpu_parts_entries_device(self._h)

    def time_ms(self, which):
        return float(self._lib.ipkgpu_parts_time_ms(self._h, which)) + self.extra_ms.get(which, 0.0)

    def counts_tensor(self):
        """torch view [n_owners, slots] int32 of the device counts (keeps self alive)."""
        return _device_tensor(self.counts_ptr(), (self.n_owners, self.slots), "int32", self)

    def entries_tensor(self):
        """torch view [num_entries, 2] int32 (branch, score bits)."""
        return _device_tensor(self.entries_ptr(), (max(self.num_entries, 0), 2), "int32", self)

    def free(self):
        if self._h:
            self._lib.ipkgpu_parts_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Db:
    """One owner's shard of the phylo-k-mer database: key -> [(branch, score)] in reference order."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle
        self.num_keys = int(lib.ipkgpu_db_num_keys(handle))
        self.num_entries = int(lib.ipkgpu_db_num_entries(handle))

    def time_ms(self):
        return float(self._lib.ipkgpu_db_time_ms(self._h))

    def keys(self):
        if self.num_keys == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(self._lib.ipkgpu_db_keys(self._h), shape=(self.num_keys,))

    def key_offsets(self):
        return np.ctypeslib.as_array(self._lib.ipkgpu_db_key_offsets(self._h), shape=(self.num_keys + 1,))

    def entries(self):
        """(branches u32 [n], scores f32 [n])"""
        if self.num_entries == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.float32)
        e = np.ctypeslib.as_array(self._lib.ipkgpu_db_entries(self._h), shape=(self.num_entries, 2))
        return e[:, 0].copy(), e[:, 1].copy().view(np.float32)

    def filter_mif0(self, engine, total_num_groups, threshold):
        """MIF0 filter values + k-mer order (filter.cpp:55-119, db_builder.cpp:281-284) on the device."""
        rc = self._lib.ipkgpu_db_filter_mif0(engine._h, self._h, int(total_num_groups), C.c_float(threshold))
        if rc != 0:
            raise engine._err(rc)

    def filter_values(self, f64=False):
        n = self.num_keys
        if n == 0:
            return np.zeros(0, np.float64 if f64 else np.float32)
        p = (self._lib.ipkgpu_db_filter_values_f64 if f64 else self._lib.ipkgpu_db_filter_values)(self._h)
        if not p:
            raise IpkGpuError(1, "filter values not computed")
        return np.ctypeslib.as_array(p, shape=(n,))

    def filter_order(self):
        n = self.num_keys
        if n == 0:
            return np.zeros(0, np.uint32)
        p = self._lib.ipkgpu_db_filter_order(self._h)
        if not p:
            raise IpkGpuError(1, "filter order not computed")
        return np.ctypeslib.as_array(p, shape=(n,))

    def filter_time_ms(self):
        return float(self._lib.ipkgpu_db_filter_time_ms(self._h))

    def keys_device_ptr(self):
        return self._lib.ipkgpu_db_keys_device(self._h)

    def key_offsets_device_ptr(self):
        return self._lib.ipkgpu_db_key_offsets_device(self._h)

    def entries_device_ptr(self):
        return self._lib.ipkgpu_db_entries_device(self._h)

    def free(self):
        if self._h:
            self._lib.ipkgpu_db_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _device_tensor(ptr, shape, dtype, owner):
    """Zero-copy torch view of engine-owned device memory via __cuda_array_interface__."""
    import torch

    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=getattr(torch, dtype), device="cuda")

    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<i4", "data": (int(ptr), False),
                                  "version": 3, "strides": None}
    h._owner = owner
    t = torch.as_tensor(h, device="cuda")
    t._ipk_owner = owner
    return t


def _score_groups_keymajor(self, logp, mat_group, k, log_eps, n_owners=1, sigma=None, sites=None, n_mats=None):
    """Scoring pass with key-major, owner-split output (see include/ipkgpu.h). logp: torch CUDA tensor
    [n_mats, sites, sigma] float32 or a raw device pointer with explicit shape."""
    _bind_keymajor(self._lib)
    mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
    keep = None
    if isinstance(logp, np.ndarray):
        import torch
        keep = torch.from_numpy(np.ascontiguousarray(logp, dtype=np.float32)).cuda()
        logp = keep
    if hasattr(logp, "data_ptr"):
        if not logp.is_cuda or not logp.is_contiguous() or logp.dtype.itemsize != 4:
            raise ValueError("device path needs a contiguous float32 CUDA tensor")
        n_mats, sites, sigma = logp.shape
        ptr = logp.data_ptr() if n_mats else 0
        import torch
        st = torch.cuda.current_stream()
        if not st.query():                     # whatever produced logp on torch's stream must be done (the engine has its own stream)
            st.synchronize()
    else:
        ptr = int(logp)
    out = C.c_void_p()
    rc = self._lib.ipkgpu_score_groups_keymajor_device(self._h, C.c_void_p(ptr), n_mats, sites, sigma,
                                                       mat_group.ctypes.data_as(C.POINTER(C.c_uint32)), k,
                                                       C.c_float(log_eps), n_owners, C.byref(out))
    del keep
    if rc != 0:
        raise self._err(rc)
    return self._adopt(Parts(self._lib, out))


def _merge_parts(self, sigma, k, owner, n_owners, counts, entries, source_offsets):
    """counts: device tensor/pointer u32 [n_sources, slots]; entries: device tensor/pointer [n, 2];
    source_offsets: entry offset of each source's block inside `entries` (host, length n_sources)."""
    _bind_keymajor(self._lib)
    so = np.ascontiguousarray(source_offsets, dtype=np.uint64)
    cp = counts.data_ptr() if hasattr(counts, "data_ptr") else int(counts)
    ep = entries.data_ptr() if hasattr(entries, "data_ptr") else int(entries or 0)
    out = C.c_void_p()
    rc = self._lib.ipkgpu_merge_parts(self._h, sigma, k, owner, n_owners, len(so), C.c_void_p(cp), C.c_void_p(ep),
                                      so.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(out))
    if rc != 0:
        raise self._err(rc)
    return self._adopt(Db(self._lib, out))


def _merge_parts_ptrs(self, sigma, k, owner, n_owners, counts_ptrs, entries_ptrs):
    """One (counts row, entry block) device pointer pair per source, in source order."""
    _bind_keymajor(self._lib)
    n = len(counts_ptrs)
    cp = (C.c_void_p * n)(*[int(x) for x in counts_ptrs])
    ep = (C.c_void_p * n)(*[int(x) for x in entries_ptrs])
    out = C.c_void_p()
    rc = self._lib.ipkgpu_merge_parts_ptrs(self._h, sigma, k, owner, n_owners, n, cp, ep, C.byref(out))
    if rc != 0:
        raise self._err(rc)
    return self._adopt(Db(self._lib, out))


def _comm_unique_id(self):
    _bind_keymajor(self._lib)
    buf = (C.c_uint8 * 128)()
    rc = self._lib.ipkgpu_comm_unique_id(buf)
    if rc != 0:
        raise IpkGpuError(rc, self._lib.ipkgpu_last_error(None).decode())
    return bytes(buf)


def _comm_prepare(self, world):
    """The local half of the communicator set-up (RCCL loaded, exchange stream and size buffers allocated): everything that
    can fail on one rank alone, done before the collective comm_init so that the ranks can agree on it first."""
    _bind_keymajor(self._lib)
    rc = self._lib.ipkgpu_comm_prepare(self._h, world)
    if rc != 0:
        raise self._err(rc)


def _comm_world_seen(self):
    """World size of the library's RCCL communicator (1 without one): what RCCL itself was initialised with."""
    _bind_keymajor(self._lib)
    return int(self._lib.ipkgpu_comm_world(self._h))


def _comm_init(self, unique_id, rank, world):
    """RCCL communicator of this context (the id's 128 bytes come from rank 0's comm_unique_id)."""
    _bind_keymajor(self._lib)
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    rc = self._lib.ipkgpu_comm_init(self._h, buf, rank, world)
    if rc != 0:
        raise self._err(rc)
    self.comm_world, self.comm_rank = world, rank


def _exchange_begin(self, parts):
    out = C.c_void_p()
    rc = self._lib.ipkgpu_exchange_begin(self._h, parts._h, C.byref(out))
    if rc != 0:
        raise self._err(rc)
    return out


def _exchange_merge(self, xfers, sigma, k):
    """-> (Db, exposed transfer time in ms); frees the transfer handles."""
    n = len(xfers)
    arr = (C.c_void_p * n)(*[x.value for x in xfers])
    out = C.c_void_p()
    exposed = C.c_double(0.0)
    rc = self._lib.ipkgpu_exchange_merge(self._h, arr, n, sigma, k, C.byref(out), C.byref(exposed))
    for x in xfers:
        self._lib.ipkgpu_xfer_free(x)
    if rc != 0:
        raise self._err(rc)
    return self._adopt(Db(self._lib, out)), float(exposed.value)


def _db_from_parts(self, parts, sigma, k):
    """Single-owner parts -> database without copying the entries (they move into the Db)."""
    _bind_keymajor(self._lib)
    out = C.c_void_p()
    rc = self._lib.ipkgpu_db_from_parts(self._h, parts._h, sigma, k, C.byref(out))
    if rc != 0:
        raise self._err(rc)
    parts.owner_offsets = parts.owner_offsets.copy()
    return self._adopt(Db(self._lib, out))


Engine.db_from_parts = _db_from_parts
Engine.merge_parts_ptrs = _merge_parts_ptrs
Engine.comm_unique_id = _comm_unique_id
Engine.comm_init = _comm_init
Engine.comm_prepare = _comm_prepare
Engine.comm_world_seen = _comm_world_seen
Engine.exchange_begin = _exchange_begin
Engine.exchange_merge = _exchange_merge
Engine.comm_world = 1
Engine.comm_rank = 0
Engine.score_groups_keymajor = _score_groups_keymajor
Engine.merge_parts = _merge_parts
