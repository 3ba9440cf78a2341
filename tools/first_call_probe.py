"""What the first scoring call of a process costs (diagnostics): engine creation, then three equal calls, wall time each.
usage: python tools/first_call_probe.py [cfg=cfg2] [groups=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipk_amd
from ipk_amd import distributed as D
from ipk_amd.synth import synth_matrices, CONFIGS
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = CONFIGS[cfgname]
n = ng * 2
mats = torch.from_numpy(np.concatenate([synth_matrices(min(250, n - i), cfg["sites"], cfg["sigma"], cfg["alpha"], cfg["seed"], first_mat=i) for i in range(0, n, 250)])).cuda()
torch.cuda.synchronize()
groups = np.repeat(np.arange(ng, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(cfg["omega"], cfg["sigma"], cfg["k"])
t = time.perf_counter(); eng = ipk_amd.Engine(0); print("Engine(0): %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
for i in range(3):
    t = time.perf_counter()
    db, parts = D.build_db_shard(eng, mats, groups, cfg["k"], eps, cfg["sigma"])
    print("call %d: %.1f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
    db.free(); parts.free()
