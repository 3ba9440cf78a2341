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
