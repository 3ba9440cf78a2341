"""numpy restatement of the key-major database assembly -- TEST INFRASTRUCTURE ONLY.

Follows db_builder.cpp:685-694: after each group, `_phylo_kmer_db.unsafe_insert(kmer, {branch, score})`
appends to the k-mer's entry list, so a key's entries are in group order.  Also provides numpy
stand-ins for the device pack/merge steps so that the multi-rank host logic
(ipk_amd/distributed.py) can be exercised with gloo on CPU.
"""
import numpy as np


def dense_code(keys, sigma, k):
    """IPK bit-packed code -> dense base-sigma value (identity for DNA)."""
    keys = np.asarray(keys, dtype=np.uint64)
    if sigma == 4:
        return keys
    out = np.zeros_like(keys)
    mul = np.uint64(1)
    for d in range(k):
        out += ((keys >> np.uint64(5 * d)) & np.uint64(31)) * mul
        mul *= np.uint64(sigma)
    return out


def build_db(group_results):
    """group_results: list of (branch_id, keys, scores) in group order -> {key: [(branch, score_bits)]}"""
    db = {}
    for branch, keys, scores in group_results:
        bits = np.asarray(scores, dtype=np.float32).view(np.uint32)
        for key, sb in zip(np.asarray(keys).tolist(), bits.tolist()):
            db.setdefault(key, []).append((int(branch), sb))
    return db


def db_shard_arrays(db, sigma, k, owner, world):
    """Owner's shard as (keys asc, key_offsets, branches, score_bits)."""
    allk = np.fromiter(db.keys(), dtype=np.uint64, count=len(db))
    dense = dense_code(allk, sigma, k)
    mine = (dense % np.uint64(world)) == np.uint64(owner)
    # ascending dense code == ascending packed code (both are lexicographic in the symbols)
    keys = allk[mine][np.argsort(dense[mine], kind="stable")].tolist()
    off, br, sc = [0], [], []
    for x in keys:
        for b, s in db[x]:
            br.append(b); sc.append(s)
        off.append(len(br))
    return (np.array(keys, dtype=np.uint32), np.array(off, dtype=np.uint64),
            np.array(br, dtype=np.uint32), np.array(sc, dtype=np.uint32))


def np_parts(group_results, sigma, k, world):
    """Stand-in for ipkgpu_score_groups_keymajor_device's output, from per-group oracle results:
    counts [world, slots] int32, entries [n, 2] int32 (owner-major, key asc, group order), owner_offsets."""
    T = sigma ** k
    slots = (T + world - 1) // world
    db = {}
    for branch, keys, scores in group_results:
        bits = np.asarray(scores, dtype=np.float32).view(np.uint32)
        for d, sb in zip(dense_code(keys, sigma, k).tolist(), bits.tolist()):
            db.setdefault(d, []).append((int(branch), sb))
    counts = np.zeros((world, slots), dtype=np.int32)
    blocks = [[] for _ in range(world)]
    for d in sorted(db):
        o, q = d % world, d // world
        counts[o, q] = len(db[d])
        blocks[o].extend(db[d])
    owner_offsets = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.uint64)
    flat = [e for b in blocks for e in b]
    entries = np.array(flat, dtype=np.uint32).reshape(-1, 2).view(np.int32)
    return counts, entries, owner_offsets


def np_merge(recv_counts, recv_entries, source_offsets, sigma, k, owner, world):
    """Stand-in for ipkgpu_merge_parts: per key concatenate the sources' entries in source order."""
    S, slots = recv_counts.shape
    cur = [int(x) for x in source_offsets]
    keys, off, rows = [], [0], []
    ent = recv_entries.view(np.uint32)
    for q in range(slots):
        n_here = 0
        for s in range(S):
            c = int(recv_counts[s, q])
            if c:
                rows.append(ent[cur[s]:cur[s] + c]); cur[s] += c; n_here += c
        if n_here:
            keys.append(q * world + owner); off.append(off[-1] + n_here)
    ent_all = np.concatenate(rows) if rows else np.zeros((0, 2), np.uint32)
    return np.array(keys, dtype=np.uint64), np.array(off, dtype=np.uint64), ent_all[:, 0].copy(), ent_all[:, 1].copy()


def pack_code(dense, sigma, k):
    dense = np.asarray(dense, dtype=np.uint64)
    if sigma == 4:
        return dense.astype(np.uint32)
    out = np.zeros_like(dense)
    for d in range(k):
        out |= (dense % np.uint64(sigma)) << np.uint64(5 * d)
        dense = dense // np.uint64(sigma)
    return out.astype(np.uint32)
