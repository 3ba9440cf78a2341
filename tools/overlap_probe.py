"""Experiment: do two halves of a workload, driven from two contexts (two streams) at once, finish sooner than one after the other?

The scoring kernel is bound by instruction issue and the reduce / key-major kernels by HBM, so their overlap is the one
whole-step gain left that needs no faster kernel.  This probe measures the best case without touching the library: two
Engine contexts on one device, each building the database of half the groups, (a) one after the other, (b) from two threads.
Usage: python tools/overlap_probe.py [--config cfg2] [--groups N] [--steps 5]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipk_amd  # noqa: E402
from ipk_amd import distributed as D  # noqa: E402
from ipk_amd.synth import synth_matrices  # noqa: E402
from ipk_amd.synth import CONFIGS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--offset-ms", type=float, default=0.0, help="start the second thread this much later")
    ap.add_argument("--phase", default="all", choices=["all", "seq", "par"], help="time only this phase (for a kernel trace of it)")
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    ng = a.groups or cfg["n_groups"]
    mpg, sites, sigma, k = cfg["mats_per_group"], cfg["sites"], cfg["sigma"], cfg["k"]
    eps = ipk_amd.log_threshold(cfg["omega"], sigma, k)
    half = ng // 2
    mats = synth_matrices(ng * mpg, sites, sigma, cfg["alpha"], cfg["seed"])
    d_all = torch.from_numpy(mats).cuda()
    d = [d_all[: half * mpg], d_all[half * mpg: 2 * half * mpg]]
    groups = [np.repeat(np.arange(0, half, dtype=np.uint32), mpg), np.repeat(np.arange(half, 2 * half, dtype=np.uint32), mpg)]
    engs = [ipk_amd.Engine(0), ipk_amd.Engine(0)]
    whole = ipk_amd.Engine(0)
    g_all = np.repeat(np.arange(0, 2 * half, dtype=np.uint32), mpg)

    def one(i):
        db, t = D.build_db_shard(engs[i], d[i], groups[i], k, eps, sigma, None, 1, 0)
        n = t.emitted
        db.free(); t.free()
        return n

    def whole_step():
        db, t = D.build_db_shard(whole, d_all[: 2 * half * mpg], g_all, k, eps, sigma, None, 1, 0)
        n = t.emitted
        db.free(); t.free()
        return n

    for _ in range(2):
        one(0); one(1); whole_step()
    torch.cuda.synchronize()
    t_whole = t_seq = float("nan")
    if a.phase == "all":
        t0 = time.perf_counter()
        for _ in range(a.steps):
            whole_step()
        t_whole = (time.perf_counter() - t0) / a.steps
    if a.phase in ("all", "seq"):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one(0); one(1)
        t_seq = (time.perf_counter() - t0) / a.steps
    if a.phase == "seq":
        print(f"{a.config} {2 * half} groups: halves one after the other {t_seq * 1e3:.2f} ms")
        return

    def worker(i, delay):
        if delay:
            time.sleep(delay)
        for _ in range(a.steps):
            one(i)

    th = [threading.Thread(target=worker, args=(0, 0.0)), threading.Thread(target=worker, args=(1, a.offset_ms * 1e-3))]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    t_par = (time.perf_counter() - t0) / a.steps
    print(f"{a.config} {2 * half} groups: whole {t_whole * 1e3:.2f} ms | halves one after the other {t_seq * 1e3:.2f} ms | "
          f"halves from two threads {t_par * 1e3:.2f} ms (offset {a.offset_ms} ms)")


if __name__ == "__main__":
    main()
