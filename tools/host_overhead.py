"""Where a build step's host time goes (diagnostics): wall time of the scoring call against its device time, and of the Python around it."""
import sys, time, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ipk_amd
from ipk_amd import engine as E
from ipk_amd.synth import synth_matrices, CONFIGS
cfgname, ng, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = CONFIGS[cfgname]
n = ng * 2
mats = torch.from_numpy(np.concatenate([synth_matrices(min(250, n - i), cfg["sites"], cfg["sigma"], cfg["alpha"], cfg["seed"], first_mat=i) for i in range(0, n, 250)])).cuda()
groups = np.repeat(np.arange(ng, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(cfg["omega"], cfg["sigma"], cfg["k"])
eng = ipk_amd.Engine(0)
acc = np.zeros(6)
for i in range(steps + 3):
    t0 = time.perf_counter()
    parts = eng.score_groups_keymajor(mats, groups, cfg["k"], eps, n_owners=1)
    t1 = time.perf_counter()
    db = eng.db_from_parts(parts, cfg["sigma"], cfg["k"])
    t2 = time.perf_counter()
    dev = parts.time_ms(E.T_TOTAL)
    t3 = time.perf_counter()
    db.free(); parts.free()
    t4 = time.perf_counter()
    if i >= 3:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, dev * 1e-3, t4 - t0]
acc /= steps
print("%s %d groups: scoring call %.1f us (device %.1f, host %.1f) | db_from_parts %.1f | one time_ms %.1f | frees %.1f | step %.1f us" % (
    cfgname, ng, acc[0] * 1e6, acc[4] * 1e6, (acc[0] - acc[4]) * 1e6, acc[1] * 1e6, acc[2] * 1e6, acc[3] * 1e6, acc[5] * 1e6))
