"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the hot path.

The expected outputs come from oracle/ipk_oracle.c (the C restatement) and are asserted equal to
oracle/np_oracle.py (the independent numpy restatement) before being written.  They are NOT outputs
of the reference binary: the reference cannot be built here (see ipk_oracle.c header), so these are
regression vectors of the oracle pair -- "parity unpinned".

Run from the repo root:  python -m oracle.gen_golden
"""
import os

import numpy as np

from ipk_amd.synth import synth_matrices
from oracle import ipk_oracle as co
from oracle import np_oracle as no

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name: sigma, k, n_groups, mats_per_group, sites, alpha, omega, seed
CASES = {
    "dna_k4": (4, 4, 2, 2, 24, 0.3, 1.5, 11),
    "dna_k7": (4, 7, 2, 2, 28, 0.2, 2.0, 12),      # the reference's own D652 test uses k=7, omega=2.0
    "dna_k8": (4, 8, 2, 2, 32, 0.1, 1.5, 13),
    "dna_k10": (4, 10, 2, 2, 40, 0.05, 1.5, 14),
    "dna_k12": (4, 12, 1, 2, 30, 0.05, 1.5, 15),
    "aa_k3": (20, 3, 2, 2, 20, 0.05, 1.5, 16),
    "aa_k4": (20, 4, 2, 1, 16, 0.03, 10.0, 17),    # the reference's own D140 test uses k=4, omega=10
    "aa_k6": (20, 6, 1, 2, 12, 0.03, 1.5, 18),
}
# cases small enough for the dense numpy enumeration (sigma^k candidates per window)
NUMPY_OK = {"dna_k4", "dna_k7", "dna_k8", "dna_k10", "aa_k3", "aa_k4"}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (sigma, k, ng, mpg, sites, alpha, omega, seed) in CASES.items():
        mats = synth_matrices(ng * mpg, sites, sigma, alpha, seed)
        eps = np.float32(co.log_threshold(omega, sigma, k))
        bits = co.bits(sigma)
        out = {"logp": mats, "sigma": sigma, "k": k, "eps": eps, "omega": np.float32(omega),
               "mats_per_group": mpg, "n_groups": ng}
        for g in range(ng):
            grp = mats[g * mpg:(g + 1) * mpg]
            keys, scores, emitted = co.explore_group(grp, k, float(eps))
            if name in NUMPY_OK:
                k2, s2, e2 = no.explore_group(grp, k, float(eps), bits)
                assert np.array_equal(keys, k2) and np.array_equal(scores.view(np.uint32), s2.view(np.uint32))
                assert emitted == e2
            out[f"keys_{g}"] = keys
            out[f"score_bits_{g}"] = scores.view(np.uint32)
            out[f"emitted_{g}"] = np.uint64(emitted)
        # one window-level vector per case (window start 1 of matrix 0)
        wk, ws = co.window(mats[0], k, 1, float(eps))
        out["win_keys"] = wk
        out["win_score_bits"] = ws.view(np.uint32)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, {g: int(out[f"emitted_{g}"]) for g in range(ng)}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
