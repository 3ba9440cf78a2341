"""Per-step phase timings of the key-major build (diagnostics)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ipk_amd
from ipk_amd import engine as E, distributed as D
from ipk_amd.synth import synth_matrices, CONFIGS
cfgname, ng, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = CONFIGS[cfgname]
n = ng * 2
mats = torch.from_numpy(np.concatenate([synth_matrices(min(250, n - i), cfg["sites"], cfg["sigma"], cfg["alpha"], cfg["seed"], first_mat=i) for i in range(0, n, 250)])).cuda()
groups = np.repeat(np.arange(ng, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(cfg["omega"], cfg["sigma"], cfg["k"])
eng = ipk_amd.Engine(0)
for i in range(steps):
    t = time.perf_counter()
    db, parts = D.build_db_shard(eng, mats, groups, cfg["k"], eps, cfg["sigma"])
    dt = time.perf_counter() - t
    print("step %d: wall %.1f ms  total %.1f  score %.1f (main %.1f reduce %.1f)  compact %.1f  free_mem %.1f GB" % (
        i, dt * 1e3, parts.time_ms(E.T_TOTAL), parts.time_ms(E.T_SCORE), parts.time_ms(E.T_SCORE_MAIN), parts.time_ms(E.T_SCORE_REDUCE),
        parts.time_ms(E.T_COMPACT), torch.cuda.mem_get_info()[0] / 1e9))
    db.free(); parts.free()
