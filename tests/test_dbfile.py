"""CPU suite: database container round trip (row n2; format parity with i2l's .ipk is unpinned)."""
import numpy as np

from ipk_amd import dbfile
from ipk_amd.synth import synth_matrices
from oracle import db_oracle as dbo
from oracle import ipk_oracle as co


def test_round_trip(tmp_path):
    sigma, k = 4, 6
    mats = synth_matrices(6, 30, sigma, 0.2, 3)
    eps = co.log_threshold(1.5, sigma, k)
    thr = co.score_threshold(1.5, sigma, k)
    full = dbo.build_db([(10 + g,) + co.explore_group(mats[2 * g:2 * g + 2], k, eps)[:2] for g in range(3)])
    keys, off, br, sc = dbo.db_shard_arrays(full, sigma, k, 0, 1)
    fv = np.array([co.mif0(sc[int(off[i]):int(off[i + 1])].view(np.float32), 5, thr) for i in range(len(keys))], dtype=np.float32)
    order = np.lexsort((keys, fv))                    # filter value, ties by key (db_builder.cpp:284)
    path = tmp_path / "db.ipkgpu"
    dbfile.write_db(path, "DNA", [(5, 0.5), (3, 0.25), (1, 0.0)], "((A:1,B:1):1,C:2);", k, 1.5, keys, off, br,
                    sc.view(np.float32), fv, order)
    hdr, recs = dbfile.read_db(path)
    assert hdr["sequence_type"] == "DNA" and hdr["kmer_size"] == k and hdr["total_num_kmers"] == len(keys)
    assert hdr["total_num_entries"] == len(br) and hdr["newick"].startswith("((A") and hdr["tree_index"][0] == (5, 0.5)
    assert [r[0] for r in recs] == keys[order].tolist()
    assert all(recs[i][1] <= recs[i + 1][1] for i in range(len(recs) - 1))
    for r in recs[:50]:
        assert [(int(b), int(s)) for b, s in zip(r[2], r[3].view(np.uint32))] == full[r[0]]


def test_merge_of_rank_shards_equals_the_single_writer(tmp_path):
    """Several GPUs: rank r owns the k-mers with code % P == r and its own filter values; merging the shard files by
    (filter value, key) must give the file one GPU writes (dbfile.merge_shards, the role of merge_stage2)."""
    rng = np.random.default_rng(5)
    n = 500
    keys = np.sort(rng.choice(4 ** 8, size=n, replace=False)).astype(np.uint32)
    lens = rng.integers(1, 6, size=n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    br = rng.integers(0, 40, size=int(off[-1])).astype(np.uint32)
    sc = (-rng.random(int(off[-1])) * 5).astype(np.float32)
    fv = np.round(rng.normal(size=n), 1).astype(np.float32)          # coarse values: plenty of ties, both signs
    fv[::50] = 0.0; fv[25::50] = -0.0                                 # and both zeros
    order = np.argsort(dbfile.filter_sort_code(fv, keys), kind="stable")
    one = tmp_path / "one.db"
    dbfile.write_db(one, "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, keys, off, br, sc, fv, order)
    for world in (2, 3):
        paths = []
        for r in range(world):
            m = keys % world == r
            ml = lens[m]
            idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in np.nonzero(m)[0]]).astype(np.int64)
            so = np.concatenate([[0], np.cumsum(ml)]).astype(np.uint64)
            paths.append(tmp_path / f"s{world}_{r}.npz")
            dbfile.write_shard(paths[-1], keys[m], so, br[idx], sc[idx], fv[m])
        merged = tmp_path / f"merged{world}.db"
        assert dbfile.merge_shards(merged, "DNA", [(3, 0.5)], "(a,b);", 8, 1.5, paths) == (n, int(off[-1]))
        assert open(merged, "rb").read() == open(one, "rb").read()
