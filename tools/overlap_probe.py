"""Feasibility probe: do the HBM-bound passes (LDS reduce, key-major assembly) of one half of the groups overlap with
the VALU-bound scoring pass of the other half when they run on two streams?  Two engine contexts, one host thread
each, half of cfg2's groups each, staggered start; compared with one context doing all groups."""
import os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ipk_amd
from ipk_amd import distributed as D
from ipk_amd.synth import CONFIGS, synth_matrices


def main():
    cfg = CONFIGS["cfg2"]
    ng, mpg, sites, sigma, k = cfg["n_groups"], cfg["mats_per_group"], cfg["sites"], cfg["sigma"], cfg["k"]
    eps = ipk_amd.log_threshold(cfg["omega"], sigma, k)
    logp = torch.from_numpy(synth_matrices(ng * mpg, sites, sigma, cfg["alpha"], cfg["seed"])).cuda()
    groups = np.repeat(np.arange(ng, dtype=np.uint32), mpg)

    def run(eng, lo, hi, reps, out, delay=0.0):
        time.sleep(delay)
        for _ in range(reps):
            db, t = D.build_db_shard(eng, logp[lo * mpg:hi * mpg], groups[lo * mpg:hi * mpg], k, eps, sigma)
            out.append(t.emitted)
            db.free(); t.free()

    one = ipk_amd.Engine(0)
    o = []
    run(one, 0, ng, 2, o)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(one, 0, ng, 6, o); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"one context, all groups: {(t1 - t0) / 6 * 1e3:.2f} ms per pass")

    a, b = ipk_amd.Engine(0), ipk_amd.Engine(0)
    oa, ob = [], []
    run(a, 0, ng // 2, 2, oa); run(b, ng // 2, ng, 2, ob)
    torch.cuda.synchronize()
    for delay in (0.0, 0.004, 0.008):
        ta = threading.Thread(target=run, args=(a, 0, ng // 2, 6, oa))
        tb = threading.Thread(target=run, args=(b, ng // 2, ng, 6, ob, delay))
        t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"two contexts, half the groups each, second delayed {delay * 1e3:.0f} ms: {(t1 - t0 - delay) / 6 * 1e3:.2f} ms per pass of all groups")
    assert oa[-1] + ob[-1] == o[-1]


if __name__ == "__main__":
    main()
