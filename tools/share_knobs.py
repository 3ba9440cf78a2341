"""A rank's share of cfg2 (125 groups) under the host's launch knobs: rounds of workgroups and chunks per wavefront and bucket.
usage: share_knobs.py [groups]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipk_amd
from ipk_amd import engine as E, distributed as D
from ipk_amd.synth import synth_matrices, CONFIGS
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 125
cfg = CONFIGS["cfg2"]
mats = torch.from_numpy(np.concatenate([synth_matrices(min(250, 2 * ng - i), cfg["sites"], 4, cfg["alpha"], cfg["seed"], first_mat=i) for i in range(0, 2 * ng, 250)])).cuda()
groups = np.repeat(np.arange(ng, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(cfg["omega"], 4, cfg["k"])
for rounds, wc in [(0, 0), (1, 1), (2, 1), (3, 1), (4, 1), (8, 1), (2, 2), (0, 0)]:
    eng = ipk_amd.Engine(0)
    eng.set_option("debug_rounds", rounds); eng.set_option("debug_wg_chunks2", wc)
    tot = {}
    for it in range(14):
        db, parts = D.build_db_shard(eng, mats, groups, cfg["k"], eps, 4)
        if it >= 4:
            for name, sel in (("total", E.T_TOTAL), ("main", E.T_SCORE_MAIN), ("reduce", E.T_SCORE_REDUCE), ("compact", E.T_COMPACT)):
                tot[name] = tot.get(name, 0.0) + parts.time_ms(sel) / 10
        db.free(); parts.free()
    print(f"rounds {rounds} wg_chunks2 {wc}: " + "  ".join(f"{k} {v:.3f}" for k, v in tot.items()), flush=True)
    eng.close()
