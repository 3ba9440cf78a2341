"""Diagnostics: the persistent reduce (default) against the workgroup-per-slice kernel (debug_flags bit 11), slice by slice."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ipk_amd
from ipk_amd.synth import synth_matrices
from oracle import ipk_oracle as co

k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sites = int(sys.argv[3]) if len(sys.argv) > 3 else 48
mats = synth_matrices(2 * n_groups, sites, 4, 0.1, 200 + k)
groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 5, 2)
eps = co.log_threshold(1.5, 4, k)
out = []
for flags in (2048, 0):
    eng = ipk_amd.Engine(0)
    eng.set_option("debug_flags", flags)
    res = eng.score_groups(mats, groups, k, eps)
    out.append([(res.group(g)[0].copy(), res.group(g)[1].copy()) for g in range(n_groups)])
    print("flags", flags, "emitted", res.emitted, "entries", res.num_entries)
    res.free(); eng.close()
TBL = 32768
for g in range(n_groups):
    (ka, sa), (kb, sb) = out[0][g], out[1][g]
    print("group", g, "old", len(ka), "new", len(kb))
    sa_, sb_ = set(ka.tolist()), set(kb.tolist())
    miss, extra = sorted(sa_ - sb_), sorted(sb_ - sa_)
    print("  missing", len(miss), miss[:10], "extra", len(extra), extra[:10])
    if len(ka) == len(kb) and np.array_equal(ka, kb):
        d = np.flatnonzero(sa.view(np.uint32) != sb.view(np.uint32))
        print("  same keys; score diffs", len(d), d[:10])
    ba = np.bincount(ka // TBL, minlength=4 ** k // TBL); bb = np.bincount(kb // TBL, minlength=4 ** k // TBL)
    bad = np.flatnonzero(ba != bb)
    print("  buckets differing", len(bad), [(int(b), int(ba[b]), int(bb[b])) for b in bad[:20]])
