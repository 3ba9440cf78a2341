"""ctypes mirror of the host-side tree code (include/ipkgpu.h, ipkgpu_tree_* / ipkgpu_ghost_plan_*).

Counterpart of what main.cpp:145,172-180 does before ipk::build: load the reference tree, extend it with ghost nodes
(extended_tree.cpp:76-162), reroot the AR tree if needed (:186-205), map extended nodes to AR nodes (ar.cpp:790-834) and
group the ghost nodes by branch (db_builder.cpp:495-553).
"""
import ctypes as C

import numpy as np

from .engine import IpkGpuError, load_library

ABI_SYMBOLS = ["ipkgpu_tree_parse", "ipkgpu_tree_load", "ipkgpu_tree_free", "ipkgpu_tree_last_error", "ipkgpu_tree_num_nodes",
               "ipkgpu_tree_num_leaves", "ipkgpu_tree_is_rooted", "ipkgpu_tree_label", "ipkgpu_tree_parent",
               "ipkgpu_tree_branch_length", "ipkgpu_tree_newick", "ipkgpu_tree_index", "ipkgpu_tree_extend", "ipkgpu_tree_reroot",
               "ipkgpu_ghost_plan_make", "ipkgpu_ghost_plan_free", "ipkgpu_ghost_plan_size", "ipkgpu_ghost_plan_ext_label",
               "ipkgpu_ghost_plan_ar_label", "ipkgpu_ghost_plan_branches"]

GHOSTS = {"both": 0, "inner-only": 1, "outer-only": 2}
_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
        L.ipkgpu_tree_parse.restype = C.c_int
        L.ipkgpu_tree_parse.argtypes = [C.c_char_p, vpp]
        L.ipkgpu_tree_load.restype = C.c_int
        L.ipkgpu_tree_load.argtypes = [C.c_char_p, vpp]
        L.ipkgpu_tree_free.restype = None
        L.ipkgpu_tree_free.argtypes = [vp]
        L.ipkgpu_tree_last_error.restype = C.c_char_p
        L.ipkgpu_tree_last_error.argtypes = []
        for n in ("ipkgpu_tree_num_nodes", "ipkgpu_tree_num_leaves"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [vp]
        L.ipkgpu_tree_is_rooted.restype = C.c_int
        L.ipkgpu_tree_is_rooted.argtypes = [vp]
        L.ipkgpu_tree_label.restype = C.c_char_p
        L.ipkgpu_tree_label.argtypes = [vp, C.c_uint32]
        L.ipkgpu_tree_parent.restype = C.c_int64
        L.ipkgpu_tree_parent.argtypes = [vp, C.c_uint32]
        L.ipkgpu_tree_branch_length.restype = C.c_double
        L.ipkgpu_tree_branch_length.argtypes = [vp, C.c_uint32]
        L.ipkgpu_tree_newick.restype = C.c_char_p
        L.ipkgpu_tree_newick.argtypes = [vp]
        L.ipkgpu_tree_index.restype = C.c_int
        L.ipkgpu_tree_index.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.ipkgpu_tree_extend.restype = C.c_int
        L.ipkgpu_tree_extend.argtypes = [vp, vpp]
        L.ipkgpu_tree_reroot.restype = C.c_int
        L.ipkgpu_tree_reroot.argtypes = [vp]
        L.ipkgpu_ghost_plan_make.restype = C.c_int
        L.ipkgpu_ghost_plan_make.argtypes = [vp, vp, vp, C.c_int, vpp]
        L.ipkgpu_ghost_plan_free.restype = None
        L.ipkgpu_ghost_plan_free.argtypes = [vp]
        L.ipkgpu_ghost_plan_size.restype = C.c_uint32
        L.ipkgpu_ghost_plan_size.argtypes = [vp]
        for n in ("ipkgpu_ghost_plan_ext_label", "ipkgpu_ghost_plan_ar_label"):
            getattr(L, n).restype = C.c_char_p
            getattr(L, n).argtypes = [vp, C.c_uint32]
        L.ipkgpu_ghost_plan_branches.restype = C.POINTER(C.c_uint32)
        L.ipkgpu_ghost_plan_branches.argtypes = [vp]
        _bound = True
    return L


def _check(L, rc):
    if rc != 0:
        raise IpkGpuError(rc, L.ipkgpu_tree_last_error().decode())


class Tree:
    def __init__(self, handle):
        self._L, self._h = _lib(), handle

    @classmethod
    def parse(cls, newick):
        L = _lib()
        h = C.c_void_p()
        _check(L, L.ipkgpu_tree_parse(newick.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        L = _lib()
        h = C.c_void_p()
        _check(L, L.ipkgpu_tree_load(str(path).encode(), C.byref(h)))
        return cls(h)

    @property
    def num_nodes(self):
        return int(self._L.ipkgpu_tree_num_nodes(self._h))

    @property
    def num_leaves(self):
        return int(self._L.ipkgpu_tree_num_leaves(self._h))

    @property
    def is_rooted(self):
        return bool(self._L.ipkgpu_tree_is_rooted(self._h))

    def label(self, postorder_id):
        return self._L.ipkgpu_tree_label(self._h, postorder_id).decode()

    def labels(self):
        return [self.label(i) for i in range(self.num_nodes)]

    def parent(self, postorder_id):
        return int(self._L.ipkgpu_tree_parent(self._h, postorder_id))

    def branch_length(self, postorder_id):
        return float(self._L.ipkgpu_tree_branch_length(self._h, postorder_id))

    def newick(self):
        return self._L.ipkgpu_tree_newick(self._h).decode()

    def index(self):
        """(num_nodes u32 [n], subtree_branch_length f64 [n]) per node in post-order (db_builder.cpp:192-197)."""
        n = self.num_nodes
        a, b = np.empty(n, np.uint32), np.empty(n, np.float64)
        _check(self._L, self._L.ipkgpu_tree_index(self._h, a.ctypes.data_as(C.POINTER(C.c_uint32)), b.ctypes.data_as(C.POINTER(C.c_double))))
        return a, b

    def extend(self):
        h = C.c_void_p()
        _check(self._L, self._L.ipkgpu_tree_extend(self._h, C.byref(h)))
        return Tree(h)

    def reroot(self):
        _check(self._L, self._L.ipkgpu_tree_reroot(self._h))

    def close(self):
        if self._h:
            self._L.ipkgpu_tree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ghost_plan(original, extended, ar_tree=None, ghosts="both"):
    """[(extended label, AR label, branch id)] in scoring order (groups first-seen, ghosts of a group in tree order)."""
    L = _lib()
    h = C.c_void_p()
    _check(L, L.ipkgpu_ghost_plan_make(original._h, extended._h, ar_tree._h if ar_tree is not None else None, GHOSTS[ghosts], C.byref(h)))
    try:
        n = int(L.ipkgpu_ghost_plan_size(h))
        br = np.ctypeslib.as_array(L.ipkgpu_ghost_plan_branches(h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return [(L.ipkgpu_ghost_plan_ext_label(h, i).decode(), L.ipkgpu_ghost_plan_ar_label(h, i).decode(), int(br[i])) for i in range(n)]
    finally:
        L.ipkgpu_ghost_plan_free(h)
