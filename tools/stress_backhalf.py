"""Randomised agreement of the two reduce kernels for 128-KB slices (debug_flags bit 11) and of the two compressed key-major
writers (bits 12 / 13): same key-major parts on random shapes.  usage: stress_backhalf.py [cases]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipk_amd
from ipk_amd.synth import synth_matrices

rng = np.random.default_rng(7)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
engs = {f: ipk_amd.Engine(0) for f in (0, 2048 | 4096, 8192)}
for f, e in engs.items():
    e.set_option("debug_flags", f)
for case in range(n_cases):
    sigma = 4 if rng.random() < 0.7 else 20
    k = int(rng.choice([11, 12])) if sigma == 4 else 6
    n_groups = int(rng.integers(1, 7 if sigma == 4 else 4))
    sites = int(rng.integers(k, 500 if sigma == 4 else 80))
    alpha = float(rng.choice([0.05, 0.1, 0.3])) if sigma == 4 else float(rng.choice([0.03, 0.06]))
    world = int(rng.choice([1, 1, 2, 3]))
    mats = synth_matrices(2 * n_groups, sites, sigma, alpha, int(rng.integers(1, 1 << 30)))
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 1, 2)
    eps = ipk_amd.log_threshold(1.5, sigma, k)
    out = []
    for f, e in engs.items():
        p = e.score_groups_keymajor(mats, groups, k, eps, n_owners=world)
        out.append((p.emitted, p.counts_tensor().cpu().numpy().copy(), p.entries_tensor().cpu().numpy().copy()))
        p.free()
    for o in out[1:]:
        assert o[0] == out[0][0] and np.array_equal(o[1], out[0][1]) and np.array_equal(o[2], out[0][2]), (case, sigma, k, n_groups, sites, alpha, world)
    print(f"case {case}: sigma {sigma} k {k} groups {n_groups} sites {sites} alpha {alpha} owners {world}: {out[0][0]} scored, {len(out[0][2])} entries ok", flush=True)
print("stress ok")
