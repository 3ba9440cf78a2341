"""Soak: many calls of every output form and scoring variant with changing shapes in one process; watches device memory.
Workspaces only grow (to the largest shape seen) and freed result blocks are cached up to a limit, so the check is a
plateau over the last quarter of the run, not a flat line from the start."""
import sys, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ipk_amd
from ipk_amd import distributed as D
from ipk_amd.synth import synth_matrices
rng = np.random.default_rng(1)
eng = ipk_amd.Engine(0)
base = None
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for it in range(n_iter):
    sigma = 4 if rng.random() < 0.7 else 20
    k = int(rng.integers(4, 13)) if sigma == 4 else int(rng.integers(2, 7))
    heavy = (sigma == 4 and k > 10) or (sigma == 20 and k > 4)
    n_mats = int(rng.integers(1, 12 if heavy else 40)); sites = int(rng.integers(k, 60 if heavy else 400))
    alpha = float(rng.choice([0.05, 0.1])) if heavy else float(rng.choice([0.05, 0.3, 1.0]))
    mats = synth_matrices(n_mats, sites, sigma, alpha, int(rng.integers(1, 1 << 30)))
    eng.set_option("variant", int(rng.choice([0, 0, 1, 3, 4])))
    groups = rng.integers(0, max(1, n_mats // 2), size=n_mats).astype(np.uint32)
    eps = ipk_amd.log_threshold(1.5, sigma, k)
    mode = it % 3
    if mode == 0:
        r = eng.score_groups(mats, groups, k, eps); r.keys(); r.free()
    elif mode == 1:
        db, parts = D.build_db_shard(eng, mats, groups, k, eps, sigma)
        db.filter_mif0(eng, 2 * n_mats + 1, ipk_amd.score_threshold(1.5, sigma, k)); db.filter_order()
        db.free(); parts.free()
    else:
        r = eng.score_groups_positions(mats, groups, k, eps); r.positions(); r.free()
    free = torch.cuda.mem_get_info()[0] / 1e9
    if it == (3 * n_iter) // 4: base = free
    if it % 25 == 0: print(f"iter {it}: free {free:.2f} GB", flush=True)
print("free at 3/4 of the run:", base, "at end:", free)
assert base - free < 2.0, "device memory keeps growing"
print("soak ok")
