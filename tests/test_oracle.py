"""CPU suite: the oracle pair (C restatement vs numpy set definition) and the golden fixtures."""
import glob
import os

import numpy as np
import pytest

from ipk_amd.synth import synth_matrices
from oracle import ipk_oracle as co
from oracle import np_oracle as no

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_log_threshold_values():
    # DNA k=10/12 and AA k=6 at omega = 1.5
    assert abs(co.log_threshold(1.5, 4, 10) - (-4.25969)) < 1e-5
    assert abs(co.log_threshold(1.5, 4, 12) - (-5.11162)) < 1e-5
    assert abs(co.log_threshold(1.5, 20, 6) - (-6.74963)) < 1e-5
    assert co.bits(4) == 2 and co.bits(20) == 5


def test_prefix_is_sequential_float_sum():
    m = synth_matrices(1, 3000, 4, 0.05, 3)[0]
    best = co.prefix_max(m)
    assert np.array_equal(best.view(np.uint32), no.prefix_max(m).view(np.uint32))
    # a pairwise/double accumulation differs in the low bits: the sequential order is part of the semantics
    dbl = np.concatenate([[0.0], np.cumsum(m.max(axis=1).astype(np.float64))]).astype(np.float32)
    assert not np.array_equal(best, dbl)
    assert np.allclose(best, dbl, rtol=1e-5)


@pytest.mark.parametrize("sigma,k,alpha", [(4, 2, 0.3), (4, 3, 0.3), (4, 5, 0.3), (4, 6, 0.1), (4, 7, 1.0),
                                           (4, 9, 0.05), (20, 2, 0.05), (20, 3, 0.05), (20, 4, 0.03)])
def test_c_oracle_matches_numpy_definition(sigma, k, alpha):
    m = synth_matrices(2, 30, sigma, alpha, 100 + k)
    eps = co.log_threshold(1.5, sigma, k)
    bits = co.bits(sigma)
    for q in range(2):
        for start in range(0, 30 - k + 1, 2):
            k1, s1 = co.window(m[q], k, start, eps)
            k2, s2 = no.window(m[q], k, start, eps, bits)
            assert np.array_equal(k1, k2)
            assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
    keys, scores, emitted = co.explore_group(m, k, eps)
    k2, s2, e2 = no.explore_group(m, k, eps, bits)
    assert emitted == e2 and np.array_equal(keys, k2)
    assert np.array_equal(scores.view(np.uint32), s2.view(np.uint32))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    mats, k, eps, mpg = z["logp"], int(z["k"]), float(z["eps"]), int(z["mats_per_group"])
    for g in range(int(z["n_groups"])):
        keys, scores, emitted = co.explore_group(mats[g * mpg:(g + 1) * mpg], k, eps)
        assert emitted == int(z[f"emitted_{g}"])
        assert np.array_equal(keys, z[f"keys_{g}"])
        assert np.array_equal(scores.view(np.uint32), z[f"score_bits_{g}"])
    wk, ws = co.window(mats[0], k, 1, eps)
    assert np.array_equal(wk, z["win_keys"]) and np.array_equal(ws.view(np.uint32), z["win_score_bits"])


def test_edge_cases():
    # zero probability -> log10 = -inf never passes the strict '>' 
    m = synth_matrices(1, 12, 4, 0.3, 5)[0]
    m[3, 1] = -np.inf
    keys, _ = co.window(m, 4, 1, -10.0)
    assert all(((int(x) >> 2) & 3) != 1 for x in keys)      # symbol 1 at window column 2 (site 3) is gone
    # an all-zero column poisons the prefix array: every later window bound is NaN -> empty result
    m2 = m.copy()
    m2[5, :] = -np.inf
    best = co.prefix_max(m2)
    assert np.isneginf(best[6:]).all()
    k2, _ = co.window(m2, 4, 7, -10.0, best)
    assert len(k2) == 0
    # put keeps the larger score; kmer_batch is key % n
    assert co.kmer_batch(1234567, 32) == 1234567 % 32


def test_mif0_matches_direct_formula():
    ls = np.array([-1.0, -2.5, -4.0], dtype=np.float32)
    N, thr = 11, np.float32(co.score_threshold(1.5, 4, 8))
    s = np.minimum(np.power(10.0, ls.astype(np.float64)), 1.0).astype(np.float32).astype(np.float64)
    S = s.sum() + (N - 3) * float(thr)
    h = lambda x: -x * np.log2(x)
    H = N * h(float(thr) / S) + sum(h(x / S) - h(float(thr) / S) for x in s)
    assert abs(co.mif0(ls, N, float(thr)) - S * (H - np.log2(N))) < 1e-12


@pytest.mark.parametrize("sigma,k", [(4, 5), (20, 2)])
def test_positions_oracle_pair(sigma, k):
    m = synth_matrices(2, 30, sigma, 0.2, 9)
    eps = co.log_threshold(1.5, sigma, k)
    k1, s1, p1, _ = co.explore_group_pos(m, k, eps)
    k2, s2, p2 = no.explore_group_pos(m, k, eps, co.bits(sigma))
    assert np.array_equal(k1, k2) and np.array_equal(s1.view(np.uint32), s2.view(np.uint32)) and np.array_equal(p1, p2)
    # positions never change the (key, score) set
    k0, s0, _ = co.explore_group(m, k, eps)
    assert np.array_equal(k0, k1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
