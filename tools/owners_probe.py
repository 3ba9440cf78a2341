"""Key-major writer time of one scoring call against the number of owners (what a rank of a P-rank build pays): diagnostics."""
import sys, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ipk_amd
from ipk_amd import engine as E
from ipk_amd.synth import synth_matrices, CONFIGS
cfgname, ng = sys.argv[1], int(sys.argv[2])
cfg = CONFIGS[cfgname]
n = ng * 2
mats = torch.from_numpy(np.concatenate([synth_matrices(min(250, n - i), cfg["sites"], cfg["sigma"], cfg["alpha"], cfg["seed"], first_mat=i) for i in range(0, n, 250)])).cuda()
groups = np.repeat(np.arange(ng, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(cfg["omega"], cfg["sigma"], cfg["k"])
eng = ipk_amd.Engine(0)
for P in (1, 2, 8):
    acc = []
    for i in range(5):
        parts = eng.score_groups_keymajor(mats, groups, cfg["k"], eps, n_owners=P)
        if i >= 2:
            acc.append((parts.time_ms(E.T_TOTAL), parts.time_ms(E.T_KM_WRITE)))
        parts.free()
    a = np.mean(acc, axis=0)
    print("%s %d groups, %d owner(s): call %.3f ms, key-major writer %.3f ms" % (cfgname, ng, P, a[0], a[1]))
