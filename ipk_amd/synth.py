"""Build-owned synthetic posterior matrices: counter-based, platform independent.

weight(seed, mat, site, state) = u^(1/alpha) + 1e-9 with u from splitmix64 of the 4-tuple;
column p = w / sum(w); the matrix stores log10(p) as float32, site-major [mat][site][state] --
the values raxmlng_reader::read_node would hand to the scoring loop (ipk/src/ar.cpp:257-260).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def synth_matrices(n_mats, sites, sigma, alpha, seed, first_mat=0):
    """Returns float32 [n_mats, sites, sigma] of log10 posteriors. Matrix index = first_mat + i."""
    with np.errstate(over="ignore"):
        mat = (np.arange(n_mats, dtype=np.uint64) + np.uint64(first_mat))[:, None, None]
        site = np.arange(sites, dtype=np.uint64)[None, :, None]
        st = np.arange(sigma, dtype=np.uint64)[None, None, :]
        ctr = _splitmix64(np.uint64(seed)) ^ (mat * np.uint64(0x100000001B3))
        ctr = _splitmix64(ctr) ^ (site * np.uint64(0x9E3779B1))
        ctr = _splitmix64(ctr) ^ st
        z = _splitmix64(ctr)
    u = ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    w = np.power(u, 1.0 / alpha) + 1e-9
    p = (w / w.sum(axis=2, keepdims=True)).astype(np.float32)
    return np.log10(p).astype(np.float32)


# BASELINE.json configs
CONFIGS = {
    "cfg2": dict(n_groups=1000, mats_per_group=2, sites=10000, sigma=4, k=10, omega=1.5, alpha=0.05, seed=42),
    "cfg3": dict(n_groups=1000, mats_per_group=2, sites=10000, sigma=4, k=12, omega=1.5, alpha=0.05, seed=42),
    # D652-shaped stand-in for configs[4] (~650 taxa -> ~1300 branch groups, ~1.4 kb alignment); the real package needs
    # git-LFS data + RAxML-ng + EPIK, none of which exist here
    "cfg5": dict(n_groups=1300, mats_per_group=2, sites=1400, sigma=4, k=10, omega=1.5, alpha=0.05, seed=44),
    "cfg4": dict(n_groups=250, mats_per_group=2, sites=3000, sigma=20, k=6, omega=1.5, alpha=0.03, seed=43),
}


def write_ancestral_probs(path, n_nodes, sites, sigma, seed=7, pool=4093):
    """A synthetic RAxML-ng `.raxml.ancestralProbs` of the benchmark's shape (header + n_nodes x sites rows
    `Node<i> TAB site TAB state TAB p_1 .. p_sigma`, 9 decimals as RAxML-ng writes with --precision 9, ar.cpp:670): what the
    loader (ipkgpu_ar_*) is timed on.  Rows are drawn from a pool of `pool` distinct probability vectors -- the text has the
    size and shape of a real file, which is all the parser's speed depends on.  Returns the file size in bytes."""
    rng = np.random.default_rng(seed)
    states = "ACGT" if sigma == 4 else "ARNDCQEGHILKMFPSTWYV"
    p = rng.dirichlet(np.full(sigma, 0.1), size=pool)
    rows = [states[int(np.argmax(v))] + "\t" + "\t".join("%.9f" % x for x in v) + "\n" for v in p]
    site_txt = [str(s + 1) + "\t" for s in range(sites)]
    with open(path, "w") as fh:
        fh.write("Node\tSite\tState\t" + "\t".join("p_" + c for c in states) + "\n")
        for i in range(n_nodes):
            lab = f"Node{i + 1}\t"
            o = (i * 7919) % pool
            fh.write("".join(lab + site_txt[s] + rows[(o + s) % pool] for s in range(sites)))
    import os
    return os.path.getsize(path)
