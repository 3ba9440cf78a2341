"""Database file writer/reader -- NOT byte-compatible with i2l's `.ipk`.

IPK streams its database through i2l::save_header / save_phylo_kmer over a Boost binary_oarchive
(ipk/src/db_builder.cpp:145-146,176-177,297-306,323-327).  Both i2l and Boost.Serialization are absent
from the reference tree, so the exact bytes cannot be reproduced or checked here ("parity unpinned").
This module writes the same LOGICAL content in the same order -- header fields as listed at
db_builder.cpp:297-305, then one record per k-mer in filter order (:323-327) -- in a plain
little-endian container of its own, and reads it back; a maintainer with i2l at hand swaps this one
module for i2l's serializer.

  magic   8s  b"IPKGPU1\\0"
  header  u32 len + sequence_type ("DNA" | "AA")                     ipk_header.sequence_type
          u64 n_index, n_index x { u32 num_nodes, f32 subtree_branch_length }   tree_index (db_builder.cpp:192-197)
          u64 len + newick of the original tree                      tree
          u64 kmer_size, f32 omega, u64 total_num_kmers, u64 total_num_entries
  k-mers  per k-mer, in filter order: u32 key, f32 filter_value, u32 n, n x { u32 branch, f32 score }
"""
import struct

import numpy as np

MAGIC = b"IPKGPU1\0"


def write_db(path, sequence_type, tree_index, newick, kmer_size, omega, keys, key_offsets, branches, scores,
             filter_values, order):
    """keys/key_offsets/branches/scores: a database shard (ascending keys); order: positions in filter order."""
    keys = np.asarray(keys, dtype=np.uint32)
    off = np.asarray(key_offsets, dtype=np.int64)
    order = np.asarray(order, dtype=np.int64)
    n_keys, n_entries = len(keys), int(off[-1]) if len(off) else 0
    lens = np.diff(off)[order]
    # word layout of the record stream: [key, fv, n, (branch, score) * n] per k-mer
    rec_words = 3 + 2 * lens
    rec_start = np.concatenate([[0], np.cumsum(rec_words)[:-1]]) if n_keys else np.zeros(0, np.int64)
    buf = np.empty(int(rec_words.sum()), dtype=np.uint32)
    buf[rec_start] = keys[order]
    buf[rec_start + 1] = np.asarray(filter_values, dtype=np.float32)[order].view(np.uint32)
    buf[rec_start + 2] = lens.astype(np.uint32)
    if n_entries:
        src = np.repeat(off[:-1][order], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
        dst = np.repeat(rec_start + 3, lens) + 2 * (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
        buf[dst] = np.asarray(branches, dtype=np.uint32)[src]
        buf[dst + 1] = np.asarray(scores, dtype=np.float32).view(np.uint32)[src]
    ti = np.zeros(len(tree_index), dtype=[("n", "<u4"), ("l", "<f4")])
    for i, (n, l) in enumerate(tree_index):
        ti[i] = (n, l)
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        st = sequence_type.encode()
        fh.write(struct.pack("<I", len(st)) + st)
        fh.write(struct.pack("<Q", len(ti)) + ti.tobytes())
        nw = newick.encode()
        fh.write(struct.pack("<Q", len(nw)) + nw)
        fh.write(struct.pack("<QfQQ", kmer_size, omega, n_keys, n_entries))
        fh.write(buf.astype("<u4").tobytes())


def read_db(path):
    """Returns (header dict, list of (key, filter_value, branches, scores) in file order)."""
    raw = open(path, "rb").read()
    assert raw[:8] == MAGIC, "not an ipk_amd database file"
    p = 8
    (n,) = struct.unpack_from("<I", raw, p); p += 4
    st = raw[p:p + n].decode(); p += n
    (ni,) = struct.unpack_from("<Q", raw, p); p += 8
    ti = np.frombuffer(raw, dtype=[("n", "<u4"), ("l", "<f4")], count=ni, offset=p); p += ni * 8
    (n,) = struct.unpack_from("<Q", raw, p); p += 8
    newick = raw[p:p + n].decode(); p += n
    k, omega, nk, ne = struct.unpack_from("<QfQQ", raw, p); p += struct.calcsize("<QfQQ")
    words = np.frombuffer(raw, dtype="<u4", offset=p)
    recs, q = [], 0
    for _ in range(nk):
        key, fvb, m = int(words[q]), words[q + 1:q + 2], int(words[q + 2])
        body = words[q + 3:q + 3 + 2 * m].reshape(m, 2)
        recs.append((key, float(fvb.view(np.float32)[0]), body[:, 0].copy(), body[:, 1].copy().view(np.float32)))
        q += 3 + 2 * m
    assert q == len(words)
    hdr = dict(sequence_type=st, tree_index=[(int(a), float(b)) for a, b in ti], newick=newick, kmer_size=k,
               omega=omega, total_num_kmers=nk, total_num_entries=ne)
    return hdr, recs
