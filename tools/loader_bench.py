#!/usr/bin/env python3
"""Loader throughput at benchmark scale ("reported separately"): synthesises a `.raxml.ancestralProbs` of a
config's shape under /tmp (cfg2: 2000 nodes x 10 000 sites x 4 states, ~1.2 GB of text), times ipkgpu_ar_open (index) and
ipkgpu_ar_read_nodes at 1 thread and at all cores, prints one JSON line.

usage: loader_bench.py [--config cfg2] [--nodes N] [--out profiles/r03_loader_cfg2.json]
Reference for the path: raxmlng_reader::read_node, ipk/src/ar.cpp:144-270 (inside the reference's stage-1 timer).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(n_nodes, sites, sigma, threads=(1, 0)):
    import ipk_amd
    from ipk_amd.loader import AncestralProbs
    from ipk_amd.synth import write_ancestral_probs
    ipk_amd.load_library()
    d = tempfile.mkdtemp(prefix="ipk_loader_")
    path = os.path.join(d, "bench.raxml.ancestralProbs")
    try:
        t = time.perf_counter(); size = write_ancestral_probs(path, n_nodes, sites, sigma); t_gen = time.perf_counter() - t
        t = time.perf_counter(); ar = AncestralProbs(path, sigma); t_open = time.perf_counter() - t
        out = {"nodes": n_nodes, "sites": sites, "sigma": sigma, "file_bytes": size, "generate_s": t_gen,
               "open_index_s": t_open, "open_index_GBps": size / t_open / 1e9, "read": []}
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        for nt in threads:
            n = nt if nt > 0 else min(cores, 16)
            best = None
            for _ in range(2):
                t = time.perf_counter(); m = ar.read(None, n); dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            out["read"].append({"threads": n, "s": best, "text_GBps": size / best / 1e9, "Mfloats_per_s": m.size / best / 1e6})
            del m
        ar.close()
        return out
    finally:
        try:
            os.remove(path); os.rmdir(d)
        except OSError:
            pass


def main():
    from ipk_amd.synth import CONFIGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    n = a.nodes or c["n_groups"] * c["mats_per_group"]
    r = measure(n, c["sites"], c["sigma"])
    r["workload"] = a.config
    line = json.dumps(r)
    print(line)
    if a.out:
        open(a.out, "w").write(json.dumps(r, indent=1) + "\n")


if __name__ == "__main__":
    main()
