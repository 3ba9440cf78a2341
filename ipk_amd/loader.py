"""ctypes mirror of the RAxML-ng ancestral-probabilities loader (include/ipkgpu.h, ipkgpu_ar_*).

Counterpart of ipk::proba_matrix / raxmlng_reader (ipk/src/proba_matrix.cpp:31-40, ipk/src/ar.cpp:144-270):
`AncestralProbs.read(labels)` returns the [n, sites, sigma] float32 log10 matrices the scoring engine takes.
"""
import ctypes as C

import numpy as np

from .engine import IpkGpuError, load_library

_bound = False

ABI_SYMBOLS = ["ipkgpu_ar_open", "ipkgpu_ar_close", "ipkgpu_ar_last_error", "ipkgpu_ar_num_nodes", "ipkgpu_ar_sites",
               "ipkgpu_ar_node_label", "ipkgpu_ar_find", "ipkgpu_ar_read_nodes"]


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        L.ipkgpu_ar_open.restype = C.c_int
        L.ipkgpu_ar_open.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.ipkgpu_ar_close.restype = None
        L.ipkgpu_ar_close.argtypes = [C.c_void_p]
        L.ipkgpu_ar_last_error.restype = C.c_char_p
        L.ipkgpu_ar_last_error.argtypes = []
        L.ipkgpu_ar_num_nodes.restype = C.c_uint32
        L.ipkgpu_ar_num_nodes.argtypes = [C.c_void_p]
        L.ipkgpu_ar_sites.restype = C.c_uint32
        L.ipkgpu_ar_sites.argtypes = [C.c_void_p]
        L.ipkgpu_ar_node_label.restype = C.c_char_p
        L.ipkgpu_ar_node_label.argtypes = [C.c_void_p, C.c_uint32]
        L.ipkgpu_ar_find.restype = C.c_int64
        L.ipkgpu_ar_find.argtypes = [C.c_void_p, C.c_char_p]
        L.ipkgpu_ar_read_nodes.restype = C.c_int
        L.ipkgpu_ar_read_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_float), C.c_uint32]
        _bound = True
    return L


class AncestralProbs:
    def __init__(self, path, sigma):
        self._L = _lib()
        h = C.c_void_p()
        rc = self._L.ipkgpu_ar_open(str(path).encode(), sigma, C.byref(h))
        if rc != 0:
            raise IpkGpuError(rc, self._L.ipkgpu_ar_last_error().decode())
        self._h = h
        self.sigma = sigma
        self.sites = int(self._L.ipkgpu_ar_sites(h))
        n = int(self._L.ipkgpu_ar_num_nodes(h))
        self.labels = [self._L.ipkgpu_ar_node_label(h, i).decode() for i in range(n)]

    def find(self, label):
        return int(self._L.ipkgpu_ar_find(self._h, label.encode()))

    def read(self, labels=None, n_threads=0):
        """[len(labels), sites, sigma] float32 log10 posteriors (all nodes, in file order, if labels is None)."""
        if labels is None:
            idx = np.arange(len(self.labels), dtype=np.uint32)
        else:
            idx = np.empty(len(labels), dtype=np.uint32)
            for i, lab in enumerate(labels):
                j = self.find(lab)
                if j < 0:
                    raise KeyError(f"Could not read the AR matrix for the node {lab}")     # ar.cpp:264-267
                idx[i] = j
        out = np.empty((len(idx), self.sites, self.sigma), dtype=np.float32)
        rc = self._L.ipkgpu_ar_read_nodes(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx),
                                          out.ctypes.data_as(C.POINTER(C.c_float)), n_threads)
        if rc != 0:
            raise IpkGpuError(rc, self._L.ipkgpu_ar_last_error().decode())
        return out

    def close(self):
        if self._h:
            self._L.ipkgpu_ar_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
