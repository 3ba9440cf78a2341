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
        rc, re_, so = D.exchange_parts(torch.from_numpy(counts), torch.from_numpy(entries), owner_off, dist, world)
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
