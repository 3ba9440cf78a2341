"""Multi-GPU host logic: shard branch groups over ranks, k-mer-keyed exchange of database parts.

The reference processes branch groups sequentially in one thread (ipk/src/db_builder.cpp:602-606; the
OpenMP pragma over groups is commented out).  Groups are independent, so rank r scores a contiguous
range of groups with no data-path collective.  The one real exchange is the k-mer-keyed merge of
the per-rank partial databases (owner = code % world, the kmer_batch rule of
branch_group.cpp:104-107): an all-to-all of per-owner blocks, after which every rank holds the
complete entry lists of the keys it owns, in global group order (= the reference's append order).

torch.distributed is plumbing only: RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
import numpy as np


def shard_range(n_groups, world, rank):
    """Contiguous, balanced range [g0, g1) of branch groups owned by `rank`."""
    base, extra = divmod(n_groups, world)
    g0 = rank * base + min(rank, extra)
    return g0, g0 + base + (1 if rank < extra else 0)


def exchange_parts(counts, entries, owner_offsets, dist, world):
    """All-to-all of the per-owner blocks.

    counts   tensor [world, slots] int32 -- row o goes to rank o
    entries  tensor [n, 2] int32         -- rows owner_offsets[o]:owner_offsets[o+1] go to rank o
    Returns (recv_counts [world, slots], recv_entries [m, 2], source_offsets [world] uint64):
    block s of the result came from rank s.
    """
    import torch

    dev = counts.device
    send_sizes = [int(owner_offsets[o + 1] - owner_offsets[o]) for o in range(world)]
    if world == 1:
        return counts, entries, np.zeros(1, dtype=np.uint64)
    if dev.type == "cuda" and dist.get_backend() == "gloo":
        # rehearsal / test transport: gloo has no device all-to-all, stage through host memory
        rc, re_, so = exchange_parts(counts.cpu(), entries.cpu(), owner_offsets, dist, world)
        return rc.to(dev), re_.to(dev), so
    ss = torch.tensor(send_sizes, dtype=torch.int64, device=dev)
    rs = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rs, ss)
    recv_sizes = [int(x) for x in rs.tolist()]
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts.contiguous())
    recv_entries = torch.empty((sum(recv_sizes), 2), dtype=entries.dtype, device=dev)
    dist.all_to_all_single(recv_entries, entries.contiguous(), output_split_sizes=recv_sizes,
                           input_split_sizes=send_sizes)
    source_offsets = np.concatenate([[0], np.cumsum(recv_sizes)[:-1]]).astype(np.uint64)
    return recv_counts, recv_entries, source_offsets


def _split_sizes(owner_offsets, world):
    return [int(owner_offsets[o + 1] - owner_offsets[o]) for o in range(world)]


class _Done:
    def wait(self):
        return True


def _a2a(dist, out, inp, out_splits=None, in_splits=None, async_op=False):
    """all_to_all_single; with the gloo rehearsal transport device tensors are staged through host memory."""
    if inp.device.type == "cuda" and dist.get_backend() == "gloo":
        o = out.cpu()
        dist.all_to_all_single(o, inp.cpu().contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(o)
        return _Done()
    w = dist.all_to_all_single(out, inp.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               async_op=async_op)
    return w if async_op else _Done()


def build_db_shard(engine, logp, mat_group, k, log_eps, sigma, dist=None, world=1, rank=0, overlap=True):
    """Scores this rank's groups and returns (this rank's database shard, parts) -- the state
    `_phylo_kmer_db` has after explore_kmers (db_builder.cpp:576-627), sharded by k-mer owner.

    With several ranks the groups are scored in two halves so that the all-to-all of the first half's
    blocks (RCCL, its own stream) runs while the second half is being scored; the merge then takes
    2 x world sources in the order (rank 0 first half, rank 0 second half, rank 1 first half, ...),
    which is global group order."""
    if world == 1:
        parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=1)
        return engine.db_from_parts(parts, sigma, k), parts

    import torch
    mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
    order = list(dict.fromkeys(mat_group.tolist()))                      # groups in first-seen order
    first = set(order[:len(order) // 2])
    in_a = np.array([g in first for g in mat_group.tolist()])
    na = int(in_a.sum())
    two_halves = (overlap and len(order) >= 2 and hasattr(logp, "data_ptr")
                  and bool(np.all(in_a[:na])) and not bool(np.any(in_a[na:])))      # halves must not interleave
    if not two_halves:
        parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=world)
        rc, re_, so = exchange_parts(parts.counts_tensor(), parts.entries_tensor(), parts.owner_offsets, dist, world)
        torch.cuda.current_stream().synchronize()
        return engine.merge_parts(sigma, k, rank, world, rc, re_, so), parts

    dev = logp.device

    def exchange(parts):
        """Starts the transfer of one half's blocks; returns the receive buffers and the pending work."""
        counts, entries = parts.counts_tensor(), parts.entries_tensor()
        send = _split_sizes(parts.owner_offsets, world)
        rs = torch.empty(world, dtype=torch.int64, device=dev)
        _a2a(dist, rs, torch.tensor(send, dtype=torch.int64, device=dev))
        recv = [int(x) for x in rs.tolist()]
        rcounts = torch.empty_like(counts)
        rentries = torch.empty((sum(recv), 2), dtype=torch.int32, device=dev)
        works = [_a2a(dist, rcounts, counts, async_op=True),
                 _a2a(dist, rentries, entries, recv, send, async_op=True)]
        return dict(recv=recv, rcounts=rcounts, rentries=rentries, works=works, keep=(counts, entries))

    pa = engine.score_groups_keymajor(logp[:na], mat_group[:na], k, log_eps, n_owners=world)
    xa = exchange(pa)                                                    # in flight while the second half is scored
    pb = engine.score_groups_keymajor(logp[na:], mat_group[na:], k, log_eps, n_owners=world)
    xb = exchange(pb)
    for w in xa["works"] + xb["works"]:
        w.wait()
    torch.cuda.current_stream().synchronize()
    # sources in global group order: (rank r, first half), (rank r, second half)
    counts = torch.stack([xa["rcounts"], xb["rcounts"]], dim=1).reshape(2 * world, -1).contiguous()
    # both receive buffers are addressed from the lower of the two base pointers (entries are 8 bytes)
    pa_ptr, pb_ptr = xa["rentries"].data_ptr(), xb["rentries"].data_ptr()
    base = min(pa_ptr, pb_ptr) if xa["rentries"].numel() and xb["rentries"].numel() else (pa_ptr or pb_ptr)
    offa = np.concatenate([[0], np.cumsum(xa["recv"])[:-1]]) + (pa_ptr - base) // 8 if xa["rentries"].numel() else np.zeros(world)
    offb = np.concatenate([[0], np.cumsum(xb["recv"])[:-1]]) + (pb_ptr - base) // 8 if xb["rentries"].numel() else np.zeros(world)
    so = np.stack([offa, offb], axis=1).reshape(-1).astype(np.uint64)
    torch.cuda.synchronize()
    db = engine.merge_parts(sigma, k, rank, world, counts, base, so)
    pa.emitted += pb.emitted
    pb.free()
    return db, pa


def write_db_file(path, sequence_type, tree_index, newick, kmer_size, omega, keys, key_offsets, branches, scores,
                  filter_values, order, workdir, dist=None, world=1, rank=0):
    """Database file from the ranks' shards.  One rank: written directly in its filter order.  Several ranks: every
    rank drops its shard into workdir/shards (the reference's on-disk mode leaves its batches in workdir/hashmaps the
    same way, db_builder.cpp:460-464), rank 0 merges them by filter value (dbfile.merge_shards).
    Returns (total k-mers, total entries) on the writing rank, None elsewhere."""
    import os
    from . import dbfile
    if world == 1:
        dbfile.write_db(path, sequence_type, tree_index, newick, kmer_size, omega, keys, key_offsets, branches, scores,
                        filter_values, order)
        return len(keys), int(key_offsets[-1]) if len(key_offsets) else 0
    sdir = os.path.join(workdir, "shards")
    os.makedirs(sdir, exist_ok=True)
    mine = os.path.join(sdir, f"shard{rank}.npz")
    dbfile.write_shard(mine, keys, key_offsets, branches, scores, filter_values)
    dist.barrier()
    out = None
    if rank == 0:
        paths = [os.path.join(sdir, f"shard{r}.npz") for r in range(world)]
        out = dbfile.merge_shards(path, sequence_type, tree_index, newick, kmer_size, omega, paths)
    dist.barrier()
    os.remove(mine)
    return out
