"""GPU suite: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import glob
import os

import numpy as np
import pytest

import ipk_amd
from ipk_amd.synth import synth_matrices
from oracle import ipk_oracle as co

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def check_against_oracle(engine, mats, mat_group, k, eps, device=False):
    mats = np.ascontiguousarray(mats, dtype=np.float32)
    mat_group = np.asarray(mat_group, dtype=np.uint32)
    if device:
        import torch
        res = engine.score_groups(torch.from_numpy(mats).cuda(), mat_group, k, eps)
    else:
        res = engine.score_groups(mats, mat_group, k, eps)
    order = list(dict.fromkeys(mat_group.tolist()))               # first-seen order
    assert res.group_ids.tolist() == order
    total_emitted = 0
    for gi, gid in enumerate(order):
        keys, scores, emitted = co.explore_group(mats[mat_group == gid], k, eps)
        gk, gs = res.group(gi)
        assert np.array_equal(gk, keys), f"group {gid}: key sets differ ({len(gk)} vs {len(keys)})"
        assert np.array_equal(gs.view(np.uint32), scores.view(np.uint32)), f"group {gid}: score bits differ"
        total_emitted += emitted
    assert res.emitted == total_emitted
    res.free()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(engine, path):
    z = np.load(path)
    mats, k, eps, mpg, ng = z["logp"], int(z["k"]), float(z["eps"]), int(z["mats_per_group"]), int(z["n_groups"])
    groups = np.repeat(np.arange(ng, dtype=np.uint32) + 7, mpg)
    res = engine.score_groups(mats, groups, k, eps)
    assert res.emitted == sum(int(z[f"emitted_{g}"]) for g in range(ng))
    for g in range(ng):
        gk, gs = res.group(g)
        assert np.array_equal(gk, z[f"keys_{g}"])
        assert np.array_equal(gs.view(np.uint32), z[f"score_bits_{g}"])
    res.free()


@pytest.mark.parametrize("k", list(range(2, 13)))
def test_dna_all_k(engine, k):
    mats = synth_matrices(4, 48, 4, 0.1, 200 + k)
    check_against_oracle(engine, mats, [5, 5, 9, 9], k, co.log_threshold(1.5, 4, k))


@pytest.mark.parametrize("k,sites", [(13, 48), (14, 40), (13, 300)])
def test_dna_k13_k14(engine, k, sites):
    """DNA k = 13, 14 (round 4; the reference accepts k up to seq_traits::max_kmer_length, ipk/src/main.cpp:131): the exact-partition
    variant with compressed tables over 4^13 / 4^14 keys; per-branch sets and the key-major database against the oracle."""
    from ipk_amd import distributed as D
    mats = synth_matrices(4, sites, 4, 0.1, 1300 + k)
    groups = np.array([5, 5, 9, 9], dtype=np.uint32)
    eps = co.log_threshold(1.5, 4, k)
    check_against_oracle(engine, mats, groups, k, eps)
    ref = {}
    for gid in (5, 9):
        keys, scores, _ = co.explore_group(mats[groups == gid], k, eps)
        for kk, sc in zip(keys.tolist(), scores.view(np.uint32).tolist()):
            ref.setdefault(kk, []).append((gid, sc))
    # one batch, then a group per batch: the merge of two batches' parts walks 4^k slots a wavefront each (launched in spans: one launch
    # of 4^13 x 64 threads is refused by the runtime)
    for workspace in ((None, 1 << 20) if sites < 100 else (None,)):
        if workspace:
            engine.set_option("workspace_bytes", workspace)
        try:
            db, parts = D.build_db_shard(engine, mats, groups, k, eps, 4)
        finally:
            engine.set_option("workspace_bytes", 8 << 30)
        dk, off = db.keys(), db.key_offsets().astype(np.int64)
        br, sc = db.entries()
        assert len(dk) == len(ref) and db.num_entries == sum(len(v) for v in ref.values())
        for i in np.random.default_rng(1).integers(0, len(dk), size=2000).tolist():
            assert [(int(b), int(x)) for b, x in zip(br[off[i]:off[i + 1]], sc[off[i]:off[i + 1]].view(np.uint32))] == ref[int(dk[i])]
        db.free(); parts.free()


def test_k13_lists_beyond_the_capped_capacity_fail_loudly(engine):
    """From k = 13 the big-list kernels' half lists are capped (6144 entries; 4^7 = 16384 cannot live in LDS): flat columns, where every
    7-symbol suffix passes, must end in an error, not in a database with k-mers missing."""
    mats = np.full((2, 14, 4), np.log10(0.25), dtype=np.float32)
    with pytest.raises(ipk_amd.IpkGpuError):
        engine.score_groups(mats, np.array([1, 1], dtype=np.uint32), 13, np.float32(-8.0))   # 13 log10(0.25) = -7.83: every k-mer passes
    # and the context is still usable
    check_against_oracle(engine, synth_matrices(2, 40, 4, 0.1, 77), [3, 3], 13, co.log_threshold(1.5, 4, 13))


@pytest.mark.parametrize("sigma,k,sites", [(4, 8, 2500), (20, 3, 700)])
def test_prefix_sums_of_several_matrices_per_workgroup(engine, sigma, k, sites):
    """matrix::preprocess (window.cpp:16-27): the running sums of a call's matrices run M to a workgroup, one chain per lane (M by the matrix
    count: 2 at cfg2); every M, with a matrix count that is no multiple of it and more sites than a chunk (2048 / M) holds."""
    mats = synth_matrices(11, sites, sigma, 0.1, 4242 + sigma)
    groups = np.array([1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6], dtype=np.uint32)
    eps = co.log_threshold(1.5, sigma, k)
    try:
        for m in (1, 2, 4, 8):
            engine.set_option("debug_prefix_mats", m)
            check_against_oracle(engine, mats, groups, k, eps)
    finally:
        engine.set_option("debug_prefix_mats", 0)


@pytest.mark.parametrize("k", list(range(2, 7)))
def test_aa_all_k(engine, k):
    mats = synth_matrices(3, 20, 20, 0.03, 300 + k)
    check_against_oracle(engine, mats, [2, 2, 4], k, co.log_threshold(1.5, 20, k))


def test_group_order_and_interleaving(engine):
    # branch ids arbitrary, matrices of a group not adjacent, a single-matrix group (INNER_ONLY ghosts)
    mats = synth_matrices(7, 40, 4, 0.1, 77)
    check_against_oracle(engine, mats, [900, 3, 900, 41, 3, 41, 12], 8, co.log_threshold(1.5, 4, 8), device=True)


def test_sites_equal_k_and_ragged_tile(engine):
    mats = synth_matrices(2, 10, 4, 0.3, 5)
    check_against_oracle(engine, mats, [0, 0], 10, co.log_threshold(1.5, 4, 10))      # one window
    mats = synth_matrices(2, 64 + 9 + 3, 4, 0.1, 6)                                      # 67 windows: ragged 2nd tile
    check_against_oracle(engine, mats, [0, 1], 10, co.log_threshold(1.5, 4, 10))


def test_rejects_bad_arguments(engine):
    mats = synth_matrices(2, 8, 4, 0.3, 5)
    with pytest.raises(ipk_amd.IpkGpuError):
        engine.score_groups(mats, [0, 0], 10, -4.0)        # sites < k (unguarded in the reference; rejected here)
    with pytest.raises(ipk_amd.IpkGpuError):
        engine.score_groups(mats, [0, 0], 13, -4.0)        # k above the supported maximum
    with pytest.raises(ipk_amd.IpkGpuError):
        engine.score_groups(np.zeros((1, 8, 5), np.float32), [0], 4, -4.0)   # unsupported alphabet


def test_big_list_fallback_dna(engine):
    # a permissive threshold makes the half lists exceed the fast path's capacity (4^5 > 160, 4^6 > 384): at k = 10 those windows are
    # re-done by the big-list kernel, at k = 12 the scoring kernel takes both lists in slices (here: every slice of L against every
    # slice of R); results must not change
    mats = synth_matrices(2, 14, 4, 1.0, 9)
    check_against_oracle(engine, mats, [0, 0], 10, -9.0)
    mats = synth_matrices(1, 14, 4, 1.0, 10)
    check_against_oracle(engine, mats, [0], 12, -12.0)


def test_big_list_fallback_aa(engine):
    mats = synth_matrices(1, 8, 20, 0.3, 11)
    check_against_oracle(engine, mats, [0], 6, -9.5)


def test_zero_probabilities_and_dead_columns(engine):
    mats = synth_matrices(2, 40, 4, 0.2, 21)
    mats[0, 7, 2] = -np.inf            # log10(0)
    mats[1, 20, :] = -np.inf           # dead column: later prefix sums are -inf, bounds become NaN
    check_against_oracle(engine, mats, [0, 1], 8, co.log_threshold(1.5, 4, 8))


def test_flat_columns_ties(engine):
    # identical columns: many equal scores, exercises max-reduce ties across windows and matrices
    col = np.log10(np.array([0.4, 0.3, 0.2, 0.1], dtype=np.float32))
    mats = np.tile(col, (2, 30, 1)).astype(np.float32)
    check_against_oracle(engine, mats, [0, 0], 6, co.log_threshold(1.0, 4, 6))


def _zero_tie_group():
    """Two matrices of one group whose only scored k-mer (code 0) scores -0.0 in the first and +0.0 in the second."""
    ninf = -np.inf
    first = np.tile(np.array([-0.0, ninf, ninf, ninf], np.float32), (12, 1))
    second = np.tile(np.array([0.0, ninf, ninf, ninf], np.float32), (12, 1))
    return np.stack([first, second])


def test_negative_zero_tie(engine):
    """The ONE documented deviation from `put` (branch_group.cpp:88-101: replace only if the stored score < the new one, so of
    -0.0 then +0.0 -- equal as floats -- the reference keeps the first, -0.0): the engine's max runs on an order-preserving
    integer code in which +0.0 is above -0.0, so it keeps +0.0 whatever the order.  Equal values, one bit apart."""
    eps = co.log_threshold(1.5, 4, 6)
    for mats, oracle_bits in ((_zero_tie_group(), 0x80000000), (_zero_tie_group()[::-1].copy(), 0x00000000)):
        keys, scores, emitted = co.explore_group(mats, 6, eps)
        assert keys.tolist() == [0] and scores.view(np.uint32).tolist() == [oracle_bits]       # the oracle follows put: first wins
        res = engine.score_groups(mats, np.array([5, 5], np.uint32), 6, eps)
        gk, gs = res.group(0)
        assert gk.tolist() == [0] and res.emitted == emitted
        assert gs[0] == scores[0] and gs.view(np.uint32).tolist() == [0x00000000]               # the engine: +0.0 either way
        res.free()


def test_idempotent_and_sorted(engine):
    mats = synth_matrices(6, 500, 4, 0.05, 31)
    groups = [0, 0, 1, 1, 2, 2]
    eps = co.log_threshold(1.5, 4, 10)
    a = engine.score_groups(mats, groups, 10, eps)
    b = engine.score_groups(mats, groups, 10, eps)
    assert a.emitted == b.emitted and np.array_equal(a.offsets, b.offsets)
    assert np.array_equal(a.keys(), b.keys()) and np.array_equal(a.scores().view(np.uint32), b.scores().view(np.uint32))
    for g in range(3):
        k_, s_ = a.group(g)
        assert np.all(np.diff(k_.astype(np.int64)) > 0)          # strictly ascending: no duplicate keys
        assert np.all(s_ > eps)                                  # every kept score passed the threshold
    a.free(); b.free()


def test_workspace_batching(engine):
    # force several batches of groups (table = 4^8 * 4 B = 256 KiB per group)
    mats = synth_matrices(10, 60, 4, 0.1, 41)
    groups = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]
    engine.set_option("workspace_bytes", 2 * 4 ** 8 * 4)
    try:
        check_against_oracle(engine, mats, groups, 8, co.log_threshold(1.5, 4, 8))
    finally:
        engine.set_option("workspace_bytes", 8 << 30)


def test_full_size_matrices_sampled(engine):
    """BASELINE cfg2 shape (10 000 sites, k=10) on a few groups: bit-exact vs the oracle."""
    mats = synth_matrices(6, 10000, 4, 0.05, 42)
    check_against_oracle(engine, mats, [0, 0, 1, 1, 2, 2], 10, co.log_threshold(1.5, 4, 10), device=True)


def test_full_size_k12_sampled(engine):
    """BASELINE cfg3 shape (10 000 sites, k=12) on two groups: the row-per-lane scoring kernel, the persistent reduce of 128-KB slices and
    the compressed writers, bit-exact vs the oracle -- per-branch sets and the key-major database (entries of a k-mer in group order)."""
    from ipk_amd import distributed as D
    mats = synth_matrices(4, 10000, 4, 0.05, 4212)
    groups = np.array([7, 7, 3, 3], dtype=np.uint32)
    eps = co.log_threshold(1.5, 4, 12)
    check_against_oracle(engine, mats, groups, 12, eps, device=True)
    db, parts = D.build_db_shard(engine, mats, groups, 12, eps, 4)
    ref = {}
    for gid in (7, 3):
        keys, scores, _ = co.explore_group(mats[groups == gid], 12, eps)
        for kk, sc in zip(keys.tolist(), scores.view(np.uint32).tolist()):
            ref.setdefault(kk, []).append((gid, sc))
    dk, off = db.keys(), db.key_offsets().astype(np.int64)
    br, sc = db.entries()
    assert len(dk) == len(ref) and db.num_entries == sum(len(v) for v in ref.values())
    rng = np.random.default_rng(3)
    for i in rng.integers(0, len(dk), size=3000).tolist():
        assert [(int(b), int(x)) for b, x in zip(br[off[i]:off[i + 1]], sc[off[i]:off[i + 1]].view(np.uint32))] == ref[int(dk[i])]
    db.free(); parts.free()


def test_full_size_aa_sampled(engine):
    mats = synth_matrices(2, 3000, 20, 0.03, 43)
    check_against_oracle(engine, mats, [0, 0], 6, co.log_threshold(1.5, 20, 6), device=True)


def test_empty_results(engine):
    """A threshold nothing can pass: every group is empty, offsets stay flat, nothing is emitted."""
    mats = synth_matrices(4, 30, 4, 0.3, 77)
    res = engine.score_groups(mats, [3, 3, 8, 8], 8, 0.5)          # log scores are <= 0 < 0.5
    assert res.emitted == 0 and res.num_entries == 0 and res.offsets.tolist() == [0, 0, 0]
    assert res.group_ids.tolist() == [3, 8]
    res.free()
    from ipk_amd import distributed as D
    db, parts = D.build_db_shard(engine, mats, np.array([3, 3, 8, 8], dtype=np.uint32), 8, 0.5, 4)
    assert db.num_keys == 0 and db.num_entries == 0 and parts.emitted == 0
    db.filter_mif0(engine, 5, 0.001)
    assert len(db.filter_order()) == 0
    db.free(); parts.free()


def test_pair_pool_exhaustion_is_redone(engine):
    """The pair pool of the stream variant is sized from an estimate; when it runs out the batch is redone
    with a bigger pool.  Force that path with a tiny first pool: results must not change."""
    mats = synth_matrices(4, 600, 4, 1.0, 123)          # flat columns: ~600 pairs per window
    eps = co.log_threshold(1.5, 4, 10)
    engine.set_option("debug_pool_chunks", 40)
    try:
        check_against_oracle(engine, mats, [1, 1, 2, 2], 10, eps)
        check_against_oracle(engine, mats, [1, 1, 2, 2], 10, eps, device=True)
    finally:
        engine.set_option("debug_pool_chunks", 0)
    check_against_oracle(engine, mats, [1, 1, 2, 2], 10, eps)


def test_many_small_groups_and_single_matrix_groups(engine):
    # 300 groups of one short matrix each: many (group, segment) workgroups with almost no work
    mats = synth_matrices(300, 24, 4, 0.2, 555)
    check_against_oracle(engine, mats, np.arange(300, dtype=np.uint32) * 7 + 1, 8, co.log_threshold(1.5, 4, 8))


def test_v1_variant_still_matches(engine):
    """variant=1 (global-atomic max-reduce, the round's first kernel) stays available and bit-exact."""
    mats = synth_matrices(4, 300, 4, 0.1, 99)
    engine.set_option("variant", 1)
    try:
        check_against_oracle(engine, mats, [0, 0, 1, 1], 10, co.log_threshold(1.5, 4, 10))
    finally:
        engine.set_option("variant", 0)


@pytest.mark.parametrize("sigma,k,sites,alpha", [(4, 5, 60, 0.3), (4, 8, 120, 0.2), (4, 10, 300, 0.1), (4, 10, 200, 1.0), (4, 12, 80, 0.1),
                                                 (20, 2, 40, 0.2), (20, 4, 40, 0.05), (20, 5, 30, 0.03), (20, 6, 40, 0.03)])
@pytest.mark.parametrize("variant", [3, 4])
def test_exact_partition_variant(engine, sigma, k, sites, alpha, variant):
    """count -> scan -> write -> LDS reduce (the default for AA k=6) on every (sigma, k) it exists for; variant 3 ends in
    dense tables, variant 4 in the compressed form (occupancy bits + rank + values in place in the pool)."""
    mats = synth_matrices(5, sites, sigma, alpha, 300 + 10 * sigma + k)
    groups = np.array([3, 8, 3, 8, 1], dtype=np.uint32)
    engine.set_option("variant", variant)
    try:
        check_against_oracle(engine, mats, groups, k, co.log_threshold(1.5, sigma, k))
        check_against_oracle(engine, mats[:2], [6, 2], k, co.log_threshold(1.5, sigma, k), device=True)
    finally:
        engine.set_option("variant", 0)


def test_exact_partition_is_the_aa_k6_default_and_handles_big_lists(engine):
    """AA k=6 takes the exact-partition path by default; broader columns push half lists past the fast path's
    capacity, so the big-list windows (queued by the count pass only) are exercised too."""
    mats = synth_matrices(3, 40, 20, 0.06, 77)          # 43 of the 105 windows have a half list above 512 entries
    eps = co.log_threshold(1.5, 20, 6)
    check_against_oracle(engine, mats, [5, 5, 6], 6, eps)
    engine.set_option("variant", 1)
    try:
        check_against_oracle(engine, mats, [5, 5, 6], 6, eps)
    finally:
        engine.set_option("variant", 0)


@pytest.mark.parametrize("sigma,k,sites", [(4, 8, 90), (4, 10, 150), (20, 3, 30), (20, 6, 14)])
def test_keep_positions_variant(engine, sigma, k, sites):
    """Row a11 (ipk-aa-pos): kept score + position of the FIRST window reaching it."""
    mats = synth_matrices(5, sites, sigma, 0.1, 70 + k)
    groups = np.array([4, 9, 4, 9, 2], dtype=np.uint32)          # interleaved matrices, one single-matrix group
    eps = co.log_threshold(1.5, sigma, k)
    res = engine.score_groups_positions(mats, groups, k, eps)
    assert res.group_ids.tolist() == [4, 9, 2]
    tot = 0
    for gi, gid in enumerate([4, 9, 2]):
        keys, scores, pos, emitted = co.explore_group_pos(mats[groups == gid], k, eps)
        a, b = int(res.offsets[gi]), int(res.offsets[gi + 1])
        assert np.array_equal(res.keys()[a:b], keys)
        assert np.array_equal(res.scores()[a:b].view(np.uint32), scores.view(np.uint32))
        assert np.array_equal(res.positions()[a:b], pos)
        tot += emitted
    assert res.emitted == tot
    res.free()


def test_keep_positions_ties_keep_first_window(engine):
    col = np.log10(np.array([0.4, 0.3, 0.2, 0.1], dtype=np.float32))
    mats = np.tile(col, (2, 25, 1)).astype(np.float32)           # every window scores every k-mer identically
    res = engine.score_groups_positions(mats, [0, 0], 6, co.log_threshold(1.0, 4, 6))
    assert res.num_entries > 0 and np.all(res.positions() == 0)  # first window of the first matrix wins every tie
    res.free()


def test_randomised_small_cases(engine):
    """Many random small configurations (alphabet, k, shape, grouping, threshold, zero probabilities) in one
    process: group-major sets, scored counts and the key-major database against the oracle, bit for bit."""
    from ipk_amd import distributed as D
    from oracle import db_oracle as dbo
    rng = np.random.default_rng(int(os.environ.get("IPK_TEST_SEED", "20261003")))     # other seeds: extra sweeps by hand
    n_cases = int(os.environ.get("IPK_TEST_CASES", "80"))
    for case in range(n_cases):
        engine.set_option("variant", int(rng.choice([0, 0, 0, 1, 3, 4, 6, 7])))   # every scoring variant and table form must give the same sets
        sigma = 4 if rng.random() < 0.65 else 20
        k = int(rng.integers(2, 13)) if sigma == 4 else int(rng.integers(2, 7))
        big = (sigma == 4 and k >= 11) or (sigma == 20 and k >= 5)        # keep the CPU oracle's share small
        sites = int(rng.integers(k, k + (8 if big else 40)))
        n_mats = int(rng.integers(1, 4 if big else 7))
        alpha = float(rng.choice([0.03, 0.1] if big else [0.03, 0.1, 0.3, 1.0]))
        mats = synth_matrices(n_mats, sites, sigma, alpha, int(rng.integers(1, 10 ** 6)))
        if rng.random() < 0.3:                                    # sprinkle log10(0)
            idx = rng.integers(0, mats.size, size=3)
            mats.reshape(-1)[idx] = -np.inf
        groups = rng.integers(0, max(1, n_mats // 2 + 1), size=n_mats).astype(np.uint32) * 11 + 3
        omega = float(rng.choice([1.0, 1.5, 2.0]))
        eps = co.log_threshold(omega, sigma, k) + float(rng.choice([0.0, 0.5] if big else [0.0, -1.0, 0.5]))
        order = list(dict.fromkeys(groups.tolist()))
        res = engine.score_groups(mats, groups, k, eps)
        assert res.group_ids.tolist() == order, case
        ref, emitted = [], 0
        for gi, gid in enumerate(order):
            keys, scores, e = co.explore_group(mats[groups == gid], k, eps)
            gk, gs = res.group(gi)
            assert np.array_equal(gk, keys) and np.array_equal(gs.view(np.uint32), scores.view(np.uint32)), (case, sigma, k, sites)
            ref.append((gid, keys, scores)); emitted += e
        assert res.emitted == emitted, case
        res.free()
        if case % 3 == 0 and not big:
            world = int(rng.integers(1, 4))
            parts = engine.score_groups_keymajor(mats, groups, k, eps, n_owners=world)
            full = dbo.build_db(ref)
            for o in range(world):
                a, b = int(parts.owner_offsets[o]), int(parts.owner_offsets[o + 1])
                db = engine.merge_parts(sigma, k, o, world, parts.counts_tensor()[o:o + 1].contiguous(),
                                        parts.entries_tensor()[a:b].contiguous(), np.zeros(1, np.uint64))
                keys, off, br, sc = dbo.db_shard_arrays(full, sigma, k, o, world)
                bb, ss = db.entries()
                assert np.array_equal(db.keys(), keys) and np.array_equal(db.key_offsets(), off), case
                assert np.array_equal(bb, br) and np.array_equal(ss.view(np.uint32), sc), case
                db.free()
            parts.free()
    engine.set_option("variant", 0)


def test_large_pair_pool_matches_small_batches(engine):
    """A pair pool beyond 4 GiB (32-bit byte offsets wrap there): 300 branch groups of 10 000-site matrices in ONE batch
    against the same groups scored 25 at a time -- every group's (key, score) set must be identical."""
    n_groups, sites, k = 300, 10000, 10
    mats = synth_matrices(n_groups * 2, sites, 4, 0.05, 4242)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32), 2)
    eps = co.log_threshold(1.5, 4, k)
    big = engine.score_groups(mats, groups, k, eps)
    offsets = big.offsets.copy()
    bk, bs = big.keys().copy(), big.scores().copy()
    emitted = big.emitted
    big.free()
    assert emitted * 8 > 5 << 30, "workload too small to put the pool beyond 4 GiB"
    tot = 0
    for g0 in range(0, n_groups, 25):
        r = engine.score_groups(mats[2 * g0:2 * g0 + 50], groups[2 * g0:2 * g0 + 50], k, eps)
        tot += r.emitted
        a, b = int(offsets[g0]), int(offsets[g0 + 25])
        assert np.array_equal(r.offsets, offsets[g0:g0 + 26] - offsets[g0])
        assert np.array_equal(r.keys(), bk[a:b]) and np.array_equal(r.scores().view(np.uint32), bs[a:b].view(np.uint32))
        r.free()
    assert tot == emitted
    # and one group of the big batch against the oracle
    keys, scores, _ = co.explore_group(mats[2 * 137:2 * 137 + 2], k, eps)
    a, b = int(offsets[137]), int(offsets[138])
    assert np.array_equal(bk[a:b], keys) and np.array_equal(bs[a:b].view(np.uint32), scores.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [8, 10, 12])
def test_quad_store_window_rebasing(engine, k):
    """The quad kernel stores pairs through 32-bit offsets from a per-wavefront base chunk and moves the base when a new
    chunk lies 4 GiB past it; debug_flags bit 3 shrinks that span to 8 chunks, so every wavefront rebases many times:
    same sets and scored counts as the oracle, also when the pool is too small at first (the retry path)."""
    mats = synth_matrices(6, 700, 4, 0.1, 99 + k)
    groups = np.array([0, 0, 1, 1, 2, 2], dtype=np.uint32)
    eps = co.log_threshold(1.5, 4, k)
    ref = [co.explore_group(mats[groups == g], k, eps) for g in range(3)]
    for pool_chunks in (0, 40):
        engine.set_option("debug_flags", 8)
        engine.set_option("debug_pool_chunks", pool_chunks)
        try:
            res = engine.score_groups(mats, groups, k, eps)
        finally:
            engine.set_option("debug_flags", 0)
            engine.set_option("debug_pool_chunks", 0)
        for g in range(3):
            gk, gs = res.group(g)
            assert np.array_equal(gk, ref[g][0]) and np.array_equal(gs.view(np.uint32), ref[g][1].view(np.uint32)), (k, pool_chunks, g)
        assert res.emitted == sum(r[2] for r in ref)
        res.free()


_EXEC_ASSERT_SCRIPT = r"""
import sys, numpy as np
import ipk_amd
from ipk_amd.synth import synth_matrices
from oracle import ipk_oracle as co
eng = ipk_amd.Engine(0)
assert eng._lib.ipkgpu_debug_exec_violations(eng._h) == 0, "not an IPK_EXEC_ASSERT build"
cases = [(4, 10, 6, 300, 0.05), (4, 12, 2, 120, 0.1), (4, 8, 4, 200, 0.1), (20, 6, 4, 40, 0.03), (20, 3, 4, 60, 0.03), (4, 10, 4, 200, 1.0)]
for sigma, k, n, sites, alpha in cases:
    mats = synth_matrices(n, sites, sigma, alpha, 77 + k)
    groups = np.repeat(np.arange(n // 2, dtype=np.uint32) + 3, 2)
    eps = co.log_threshold(1.5, sigma, k)
    res = eng.score_groups(mats, groups, k, eps)            # group-major: scoring kernels + LDS reduce + CSR writers
    tot = 0
    for gi in range(n // 2):
        keys, scores, emitted = co.explore_group(mats[2 * gi:2 * gi + 2], k, eps)
        gk, gs = res.group(gi)
        assert np.array_equal(gk, keys) and np.array_equal(gs.view(np.uint32), scores.view(np.uint32))
        tot += emitted
    assert res.emitted == tot
    res.free()
    parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=1)   # key-major: km_write / km_write_c
    assert parts.emitted == tot
    parts.free()
v = eng._lib.ipkgpu_debug_exec_violations(eng._h)
print("EXEC_VIOLATIONS", v)
sys.exit(0 if v == 0 else 3)
"""


def test_exec_assert_build():
    """The inline-asm helpers that write `exec` (store8_lanes, pool_store_lanes, pool_store_inside, km_write_c's LDS scatter)
    restore it to all ones: correct only when entered with every lane enabled.  The IPK_EXEC_ASSERT build of the library
    counts entries with a partial mask; the count must stay 0 across every kernel family, results still bit-exact."""
    import subprocess
    import sys
    from ipk_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = B.build_variant("execassert")
    out = subprocess.run([sys.executable, "-c", _EXEC_ASSERT_SCRIPT], capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(os.environ, IPKGPU_LIB=lib, PYTHONPATH=root))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "EXEC_VIOLATIONS 0" in out.stdout


@pytest.mark.parametrize("sigma,k,sites,alpha,limit", [(4, 10, 300, 0.05, 20 << 20), (20, 6, 40, 0.03, 4 << 20)],
                         ids=["dna_k10_chunked_pool", "aa_k6_exact_partition"])
def test_pool_that_does_not_fit_means_smaller_batches(sigma, k, sites, alpha, limit):
    """The pair pool of a batch does not fit device memory (here: an artificial limit): the engine halves the batch and
    scores it again instead of failing -- the reference's valve for "does not fit" is its on-disk mode
    (db_builder.cpp:673-681); results are the oracle's either way, group order included."""
    n_groups = 8
    mats = synth_matrices(2 * n_groups, sites, sigma, alpha, 991)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 30, 2)
    eps = co.log_threshold(1.5, sigma, k)
    eng = ipk_amd.Engine(0)
    try:
        one = eng.score_groups(mats, groups, k, eps)
        launches_free = one.time_ms(4)                                  # IPKGPU_T_SCORE_LAUNCHES
        one.free()
        eng.set_option("debug_pool_limit_bytes", limit)
        res = eng.score_groups(mats, groups, k, eps)
        assert res.time_ms(4) > launches_free, "the limit did not force smaller batches"
        res.free()
        check_against_oracle(eng, mats, groups, k, eps)
        parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=1)
        db = eng.db_from_parts(parts, sigma, k)
        full, emitted = _oracle_db(mats, groups, k, eps)
        assert parts.emitted == emitted and parts.time_ms(4) > 1
        keys, off = db.keys(), db.key_offsets()
        br, sc = db.entries()
        assert len(keys) == len(full)
        bits = ipk_amd.bits_per_symbol(sigma)
        for i in range(0, len(keys), max(1, len(keys) // 500)):          # a sample of the keys, every entry of each
            e = full[int(keys[i])]
            assert [int(b) for b in br[off[i]:off[i + 1]]] == [b for b, _ in e]
            assert [int(x) for x in sc[off[i]:off[i + 1]].view(np.uint32)] == [x for _, x in e]
        db.free(); parts.free()
    finally:
        eng.close()


def _oracle_db(mats, groups, k, eps):
    """key -> [(branch, score bits)] in group (first-seen) order, and the scored count: the oracle's explore_group per group."""
    full, emitted = {}, 0
    for gid in dict.fromkeys(np.asarray(groups).tolist()):
        keys, scores, e = co.explore_group(mats[np.asarray(groups) == gid], k, eps)
        emitted += e
        for key, s in zip(keys.tolist(), scores.view(np.uint32).tolist()):
            full.setdefault(key, []).append((gid, s))
    return full, emitted


@pytest.mark.parametrize("sigma,k,sites,alpha", [(4, 8, 90, 0.1), (4, 10, 150, 0.05), (4, 12, 60, 0.05), (20, 4, 30, 0.03)],
                         ids=["dna_k8", "dna_k10", "dna_k12", "aa_k4"])
@pytest.mark.parametrize("variant", [6, 7], ids=["compressed", "dense"])
def test_chunk_fed_reduce_in_both_table_forms(sigma, k, sites, alpha, variant):
    """The LDS reduce over the chunked pair pool ends either in dense per-group tables or in the compressed form (occupancy
    bits + rank + score codes; the default for DNA k = 11, 12): both must give the oracle's sets, per branch and key-major."""
    n_groups = 5
    mats = synth_matrices(2 * n_groups, sites, sigma, alpha, 4100 + k)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 11, 2)
    eps = co.log_threshold(1.5, sigma, k)
    eng = ipk_amd.Engine(0)
    try:
        eng.set_option("variant", variant)
        check_against_oracle(eng, mats, groups, k, eps)
        parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=2)
        full, emitted = _oracle_db(mats, groups, k, eps)
        assert parts.emitted == emitted and parts.num_entries == sum(len(v) for v in full.values())
        for owner in range(2):
            db = eng.merge_parts_ptrs(sigma, k, owner, 2, [parts.counts_ptr() + 4 * owner * parts.slots],
                                      [parts.entries_ptr() + 8 * int(parts.owner_offsets[owner])])
            keys, off = db.keys(), db.key_offsets()
            br, sc = db.entries()
            for i in range(0, len(keys), max(1, len(keys) // 300)):
                e = full[int(keys[i])]
                assert [int(b) for b in br[off[i]:off[i + 1]]] == [b for b, _ in e]
                assert [int(x) for x in sc[off[i]:off[i + 1]].view(np.uint32)] == [x for _, x in e]
            db.free()
        parts.free()
    finally:
        eng.close()


@pytest.mark.parametrize("k,alpha", [(12, 0.35), (11, 0.3), (12, 1.0)], ids=["k12_dense_rows", "k11_dense_rows", "k12_flat"])
def test_row_per_lane_join_with_dense_rows(engine, k, alpha):
    """DNA k = 11, 12 (row-per-lane final join): flattish columns give rows with dozens of passing pairs each, so the eight
    rows of a key bucket reserve more than a chunk's worth in ONE round -- the bucket's chunk is closed early and crossed
    again in the new chunk (LaneAppender::roll_at, the re-examination loop) -- and, fully flat, lists beyond the fast
    path's capacity (taken in slices by the scoring kernel since round 4; compressed tables)."""
    mats = synth_matrices(4, 70, 4, alpha, 700 + k)
    check_against_oracle(engine, mats, [3, 3, 8, 8], k, co.log_threshold(1.5, 4, k))
    check_against_oracle(engine, mats, [3, 3, 8, 8], k, co.log_threshold(1.5, 4, k), device=True)


def _db_arrays(eng, mats, groups, k, eps, sigma):
    parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=1)
    db = eng.db_from_parts(parts, sigma, k)
    br, sc = db.entries()
    out = (parts.emitted, db.keys().copy(), db.key_offsets().copy(), br.copy(), sc.view(np.uint32).copy())
    db.free(); parts.free()
    return out


def _same_db(a, b):
    return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


@pytest.mark.parametrize("sigma,k,sites", [(4, 10, 400), (4, 12, 200), (20, 6, 40)], ids=["dna_k10", "dna_k12_compressed", "aa_k6"])
def test_calls_without_waits_match_calls_with_them(sigma, k, sites):
    """From its second call on, a context's key-major build waits on the stream ONCE: the wait after pass 1 is skipped (chunk
    index and reduce take the chunk count from the device), the scored count and the pool's state come back with the writer's
    totals, and the writer's output is allocated from the previous call's size before the total is known.  debug_flags bit 6
    switches all of that off; both ways give the same database, and the oracle's."""
    n_groups = 6
    alpha = 0.05 if sigma == 4 else 0.03
    mats = synth_matrices(2 * n_groups, sites, sigma, alpha, 4242)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 3, 2)
    eps = co.log_threshold(1.5, sigma, k)
    plain = ipk_amd.Engine(0)
    plain.set_option("debug_flags", 64)
    eng = ipk_amd.Engine(0)
    try:
        want = _db_arrays(plain, mats, groups, k, eps, sigma)
        full, emitted = _oracle_db(mats, groups, k, eps)
        assert want[0] == emitted and len(want[1]) == len(full)
        for i in range(4):                                            # call 0 calibrates; 1.. run without the waits
            got = _db_arrays(eng, mats, groups, k, eps, sigma)
            assert _same_db(got, want), "call %d differs" % i
    finally:
        eng.close(); plain.close()


def test_writer_output_estimate_that_is_too_small():
    """The key-major writer's output is allocated from the previous call's entry count; a call that produces MORE finds the
    writer returning untouched and runs it again with the true size.  Same database as a fresh context's."""
    sigma, k, sites, n_groups = 4, 10, 300, 6
    eps = co.log_threshold(1.5, sigma, k)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32), 2)
    sharp = synth_matrices(2 * n_groups, sites, sigma, 0.02, 77)       # few pairs per window
    flat = synth_matrices(2 * n_groups, sites, sigma, 0.3, 78)         # many more
    eng, fresh = ipk_amd.Engine(0), ipk_amd.Engine(0)
    try:
        a = _db_arrays(eng, sharp, groups, k, eps, sigma)
        b = _db_arrays(eng, flat, groups, k, eps, sigma)               # estimate from `sharp`: too small
        assert len(b[3]) > 1.2 * len(a[3]), "the second workload must have more entries for this test to mean anything"
        assert _same_db(b, _db_arrays(fresh, flat, groups, k, eps, sigma))
        c = _db_arrays(eng, sharp, groups, k, eps, sigma)              # and back: an estimate that is too large
        assert _same_db(c, a)
    finally:
        eng.close(); fresh.close()


def test_pool_that_runs_out_in_a_call_without_waits():
    """A call that skipped the wait after pass 1 learns that its pair pool ran out only at its last wait: the batch is scored
    again with a larger pool (debug_pool_chunks forces the small first attempt)."""
    sigma, k, sites, n_groups = 4, 10, 500, 8
    eps = co.log_threshold(1.5, sigma, k)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32), 2)
    mats = synth_matrices(2 * n_groups, sites, sigma, 0.05, 1234)
    eng, fresh = ipk_amd.Engine(0), ipk_amd.Engine(0)
    try:
        want = _db_arrays(fresh, mats, groups, k, eps, sigma)
        first = _db_arrays(eng, mats, groups, k, eps, sigma)           # calibrates; the next call skips the waits
        assert _same_db(first, want)
        eng.set_option("debug_pool_chunks", 40)
        again = _db_arrays(eng, mats, groups, k, eps, sigma)
        assert _same_db(again, want)
        eng.set_option("debug_pool_chunks", 0)
        assert _same_db(_db_arrays(eng, mats, groups, k, eps, sigma), want)
    finally:
        eng.close(); fresh.close()


@pytest.mark.parametrize("k", [8, 10, 11, 12])
def test_workgroups_that_draw_their_tiles(k):
    """The quad kernel's workgroups of a group draw their tiles from the group's counter (first tile by position) where a
    workgroup has four or more to do; debug_flags bit 8 forces that on small inputs, bit 7 forces fixed ranges.  Same sets,
    same scored counts as the oracle either way -- which workgroup scores a window does not matter."""
    sigma, sites = 4, 900
    mats = synth_matrices(6, sites, sigma, 0.08, 500 + k)
    groups = np.array([4, 4, 9, 9, 9, 2], dtype=np.uint32)
    eps = co.log_threshold(1.5, sigma, k)
    for flags in (256, 128):
        eng = ipk_amd.Engine(0)
        try:
            eng.set_option("debug_flags", flags)
            check_against_oracle(eng, mats, groups, k, eps)
            check_against_oracle(eng, mats, groups, k, eps)          # second call: pool calibrated, no wait after pass 1
        finally:
            eng.close()


@pytest.mark.parametrize("k,n_groups,sites", [(12, 3, 300), (11, 2, 500), (12, 1, 40), (12, 7, 1500)])
def test_persistent_reduce_of_128_kb_slices(k, n_groups, sites):
    """DNA k = 11, 12: a (group, key bucket) slice is a 128-KB LDS table, one workgroup per CU.  reduce_buckets_pipe_kernel keeps
    a workgroup on its CU and streams the slices' chunks through two register buffers (kernels_reduce_pipe.hpp); the
    workgroup-per-slice kernel stays behind debug_flags bit 11.  Both against the oracle (branch_group.cpp:88-101, `put`), group-major
    and key-major; more slices than CUs, workgroups with one slice and with several, empty slices (40 sites touch few buckets)."""
    sigma = 4
    mats = synth_matrices(2 * n_groups, sites, sigma, 0.07, 700 + k + n_groups)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32) + 3, 2)
    eps = co.log_threshold(1.5, sigma, k)
    parts_out = []
    for flags in (0, 2048):
        eng = ipk_amd.Engine(0)
        try:
            eng.set_option("debug_flags", flags)
            if sites <= 500:
                check_against_oracle(eng, mats, groups, k, eps)
            parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=1)
            parts_out.append((parts.emitted, parts.counts_tensor().cpu().numpy().copy(), parts.entries_tensor().cpu().numpy().copy()))
            parts.free()
        finally:
            eng.close()
    assert parts_out[0][0] == parts_out[1][0]
    assert np.array_equal(parts_out[0][1], parts_out[1][1]) and np.array_equal(parts_out[0][2], parts_out[1][2])


@pytest.mark.parametrize("n_groups,world", [(70, 1), (130, 3), (5, 2)])
def test_dense_writer_with_line_aligned_stores(n_groups, world):
    """km_write_lines_kernel cuts every store at a 128-byte line of the output and carries the rest of a key's entries as a tail of
    fewer than 16 (in registers, compacted with the next tile's entries through LDS); km_write_kernel (debug_flags bit 9) stores
    tile by tile.  Same parts from both, several tiles per key (more than 64 groups), owners whose blocks start at odd offsets,
    and a second batch appending behind the first (cursors that start in the middle of a line)."""
    sigma, k, sites = 4, 8, 60
    mats = synth_matrices(n_groups, sites, sigma, 0.6, 31 + n_groups)
    groups = np.arange(n_groups, dtype=np.uint32) + 2
    eps = co.log_threshold(1.5, sigma, k)
    out = []
    for flags in (1024, 512):                                         # line-cut stores whatever the group count | tile by tile
        eng = ipk_amd.Engine(0)
        try:
            eng.set_option("debug_flags", flags)
            eng.set_option("variant", 7)                              # dense tables whatever the occupancy
            if n_groups > 100:
                eng.set_option("workspace_bytes", 70 * (sigma ** k) * 4)   # two batches of groups: the second appends behind the first
            parts = eng.score_groups_keymajor(mats, groups, k, eps, n_owners=world)
            out.append((parts.emitted, parts.counts_tensor().cpu().numpy().copy(), parts.entries_tensor().cpu().numpy().copy(),
                        list(parts.owner_offsets)))
            parts.free()
        finally:
            eng.close()
    assert out[0][0] == out[1][0] and out[0][3] == out[1][3]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    full, emitted = _oracle_db(mats, groups, k, eps)
    assert out[0][0] == emitted and int(out[0][1].sum()) == sum(len(v) for v in full.values())


def test_randomised_cases_against_the_oracle(engine):
    """Ten random cases (tools/fuzz_parity.py, fixed seed; profiles/r04_fuzz_parity.txt holds a ten-minute run of 228): alphabet, k (DNA up to 12 here: the 4^13 / 4^14 count rows make a case take
    seconds), sites, grouping -- contiguous or interleaved --, column concentration, omega, owners, batch size, prefix-kernel shape;
    the per-branch result and every owner's database shard bit for bit as the oracle has them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, fails = mod.run(cases=10, seed=20260, eng=engine, max_k_dna=12, log=lambda m: None)
    assert n == 10 and fails == [], fails
