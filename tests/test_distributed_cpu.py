"""CPU suite: the N>1 host path (group sharding + k-mer-keyed all-to-all) with gloo, world_size 2 and 3.

The device pack/merge kernels are replaced by their numpy restatements (oracle/db_oracle.py); what is
under test is ipk_amd/distributed.py: shard ranges, split sizes, all-to-all plumbing and the ordering
contract (merged shard == shard of the single-process database, entries in global group order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ipk_amd import distributed as D
from ipk_amd.synth import synth_matrices
from oracle import db_oracle as dbo
from oracle import ipk_oracle as co


def test_shard_range_partitions():
    for n in (1, 7, 10, 1000):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, w, i) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, sigma, k, n_groups, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mpg, sites = 2, 24
        mats = synth_matrices(n_groups * mpg, sites, sigma, 0.2, 77)
        eps = co.log_threshold(1.5, sigma, k)
        g0, g1 = D.shard_range(n_groups, world, rank)
        mine = []
        for g in range(g0, g1):
            keys, scores, _ = co.explore_group(mats[g * mpg:(g + 1) * mpg], k, eps)
            mine.append((100 + g, keys, scores))
        counts, entries, owner_off = dbo.np_parts(mine, sigma, k, world)
        rc, re_, rs = D.exchange_parts(torch.from_numpy(counts), torch.from_numpy(entries), owner_off, dist, world)
        so = np.concatenate([[0], np.cumsum(rs)[:-1]]).astype(np.uint64)
        keys, off, br, sc = dbo.np_merge(rc.numpy(), re_.numpy(), so, sigma, k, rank, world)
        np.savez(os.path.join(out_dir, f"shard{rank}.npz"), keys=dbo.pack_code(keys, sigma, k), off=off, br=br, sc=sc)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sigma,k", [(2, 4, 6), (3, 4, 5), (2, 20, 3)])
def test_exchange_matches_single_process_db(tmp_path, world, sigma, k):
    n_groups = 5
    mp.spawn(_worker, args=(world, _free_port(), sigma, k, n_groups, str(tmp_path)), nprocs=world, join=True)
    mats = synth_matrices(n_groups * 2, 24, sigma, 0.2, 77)
    eps = co.log_threshold(1.5, sigma, k)
    full = dbo.build_db([(100 + g,) + co.explore_group(mats[2 * g:2 * g + 2], k, eps)[:2] for g in range(n_groups)])
    seen = 0
    for r in range(world):
        z = np.load(tmp_path / f"shard{r}.npz")
        keys, off, br, sc = dbo.db_shard_arrays(full, sigma, k, r, world)
        assert np.array_equal(z["keys"], keys)
        assert np.array_equal(z["off"], off)
        assert np.array_equal(z["br"], br) and np.array_equal(z["sc"], sc)
        seen += len(keys)
    assert seen == len(full)


def test_piece_cuts():
    g = np.array([5, 5, 9, 9, 2, 2, 7, 7], dtype=np.uint32)
    assert D.piece_cuts(g, 1) == [0, 8] and D.piece_cuts(g, 2) == [0, 4, 8] and D.piece_cuts(g, 4) == [0, 2, 4, 6, 8]
    assert D.piece_cuts(g, 3) == [0, 4, 6, 8]                       # 4 groups over 3 pieces: 2 + 1 + 1
    assert D.piece_cuts(g[:2], 4) == [0, 2, 2, 2, 2]                # fewer groups than pieces: trailing pieces empty
    assert D.piece_cuts(np.zeros(0, np.uint32), 3) == [0, 0, 0, 0]  # a rank without groups still cuts
    assert D.piece_cuts(np.array([1, 2, 1, 2]), 2) is None          # interleaved matrices cannot be sliced
    assert D.piece_cuts(np.array([1, 2, 1, 2]), 1) == [0, 4]


def _agree_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0: plenty of groups; rank 1: ONE group; rank 2: no groups at all -- all must end with the same count
        shapes = [np.repeat(np.arange(6), 2), np.array([40, 40]), np.zeros(0, np.int64)]
        a = D.agree_on_pieces(shapes[rank], 4, dist, "cpu")
        # interleaved matrices on one rank only: everybody falls back to one piece
        shapes2 = [np.repeat(np.arange(6), 2), np.array([1, 2, 1, 2]), np.repeat(np.arange(3), 2)]
        b = D.agree_on_pieces(shapes2[rank], 4, dist, "cpu")
        open(os.path.join(out_dir, f"agree{rank}.txt"), "w").write(f"{a} {b}")
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_piece_count(tmp_path):
    """Uneven shards, an empty rank and an unsliceable rank must not desynchronise the collectives (one rank issuing
    fewer exchanges than its peers would hang the job)."""
    mp.spawn(_agree_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    got = [open(tmp_path / f"agree{r}.txt").read() for r in range(3)]
    assert got == ["4 1", "4 1", "4 1"]


def test_default_pieces_follow_the_share_size():
    from ipk_amd import distributed as D
    assert [D.default_pieces(n) for n in (0, 1, 47, 48, 125, 250, 1000)] == [1, 1, 1, 1, 2, 4, 4]


def test_pieces_model_arithmetic():
    """The piece rule's model (printed into the N > 1 bench line): more pieces cost their fixed call time and hide all but the last
    piece's transfer while a piece's scoring outlasts it; one rank has nothing to exchange."""
    from ipk_amd import distributed as D
    m = D.pieces_model(125, 8, 118_000_000, 0.0197)
    by = {r["pieces"]: r for r in m["by_pieces"]}
    assert abs(by[1]["scoring_ms"] - (0.52 + 125 * 0.0197)) < 1e-9 and by[2]["extra_fixed_ms"] == 0.52
    full = 8.0 * 118_000_000 * 7 / 8 / 7 / 100e9 * 1e3                     # one link's share of the rank's entries, ms
    assert abs(by[1]["exchange_exposed_ms"] - full) < 1e-9 and abs(by[2]["exchange_exposed_ms"] - full / 2) < 1e-9
    assert by[3]["step_ms"] > by[3]["scoring_ms"]
    slow = D.pieces_model(4, 8, 118_000_000, 0.0197, link_gb_s=1.0)         # a transfer far longer than a piece's scoring stays exposed
    assert slow["by_pieces"][1]["exchange_exposed_ms"] > slow["by_pieces"][0]["exchange_exposed_ms"] * 0.9
    one = D.pieces_model(1000, 1, 10 ** 9, 0.0186)
    assert all(r["exchange_exposed_ms"] == 0.0 and r["merge_ms"] == 0.0 for r in one["by_pieces"])
