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


# ---- several GPUs: every rank owns the k-mers with code % P == rank ---------------------------------
# Filter values are per k-mer, so each rank computes them on its own shard; the final file is the merge of
# the P shards by filter value -- the role merge_stage2 plays for the reference's on-disk batches
# (db_builder.cpp:392-458: batch files opened together, smallest filter value written next).

def filter_sort_code(filter_values, keys):
    """The engine's filter order as one integer per k-mer: order-preserving code of the float32 filter value, ties by
    ascending key (ipk_amd/csrc/kernels_filter.hpp, filter_sortkey_kernel)."""
    u = np.asarray(filter_values, dtype=np.float32).view(np.uint32).astype(np.uint64)
    code = np.where(u & np.uint64(0x80000000), ~u & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))
    return (code << np.uint64(32)) | np.asarray(keys, dtype=np.uint64)


def splitmix_unit(keys):
    """A reproducible value in [0, 1) per k-mer code (the `random` filter): independent of the sharding."""
    x = np.asarray(keys, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def write_shard(path, keys, key_offsets, branches, scores, filter_values):
    """One rank's shard (ascending keys) with its filter values, as plain arrays for merge_shards."""
    with open(path, "wb") as fh:
        np.savez(fh, keys=np.asarray(keys, dtype=np.uint32), off=np.asarray(key_offsets, dtype=np.uint64),
                 br=np.asarray(branches, dtype=np.uint32), sc=np.asarray(scores, dtype=np.float32).view(np.uint32),
                 fv=np.asarray(filter_values, dtype=np.float32))


def merge_shards(path, sequence_type, tree_index, newick, kmer_size, omega, shard_paths):
    """Merges the ranks' shards by (filter value, key) into one database file, identical to the file a single
    GPU writes for the same input.  Returns (total k-mers, total entries)."""
    keys, lens, br, sc, fv = [], [], [], [], []
    for sp in shard_paths:
        z = np.load(sp)
        keys.append(z["keys"]); lens.append(np.diff(z["off"].astype(np.int64))); br.append(z["br"]); sc.append(z["sc"]); fv.append(z["fv"])
    keys = np.concatenate(keys); lens = np.concatenate(lens); fv = np.concatenate(fv)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    order = np.argsort(filter_sort_code(fv, keys), kind="stable")
    write_db(path, sequence_type, tree_index, newick, kmer_size, omega, keys, off, np.concatenate(br),
             np.concatenate(sc).view(np.float32), fv, order)
    return len(keys), int(off[-1])
