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


def _piece_cuts(mat_group, order, n):
    """Matrix cut points of n contiguous ranges of the groups (first-seen order), or None if the matrices of the
    ranges interleave (the pieces are scored from slices of the matrix array)."""
    piece_of = {g: min(n - 1, i * n // len(order)) for i, g in enumerate(order)}
    pid = np.array([piece_of[g] for g in mat_group.tolist()])
    if np.any(np.diff(pid) < 0):
        return None
    return [int(np.searchsorted(pid, j, side="left")) for j in range(n)] + [len(pid)]


def build_db_shard(engine, logp, mat_group, k, log_eps, sigma, dist=None, world=1, rank=0, overlap=True, pieces=None):
    """Scores this rank's groups and returns (this rank's database shard, parts) -- the state
    `_phylo_kmer_db` has after explore_kmers (db_builder.cpp:576-627), sharded by k-mer owner.

    With several ranks the groups are scored in `pieces` contiguous ranges (default 4, IPK_DIST_PIECES) so that the
    all-to-all of one range's blocks (RCCL, its own stream) runs while the next range is being scored -- only the last
    range's transfer is exposed; the merge then takes pieces x world sources in the order (rank 0 piece 0, rank 0
    piece 1, ..., rank 1 piece 0, ...), which is global group order.  The ranks agree on the piece count (the smallest
    any of them can do), so uneven shards cannot desynchronise the collectives."""
    if world == 1:
        parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=1)
        return engine.db_from_parts(parts, sigma, k), parts

    import os
    import torch
    mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
    order = list(dict.fromkeys(mat_group.tolist()))                      # groups in first-seen order
    want = pieces if pieces is not None else int(os.environ.get("IPK_DIST_PIECES", "4"))
    n = 1
    if overlap and hasattr(logp, "data_ptr"):
        n = max(1, min(want, len(order)))
        while n > 1 and _piece_cuts(mat_group, order, n) is None:
            n -= 1
    on_gpu = dist.get_backend() != "gloo"
    agreed = torch.tensor([n], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    n = int(agreed.item())
    if n <= 1:
        parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=world)
        rc, re_, so = exchange_parts(parts.counts_tensor(), parts.entries_tensor(), parts.owner_offsets, dist, world)
        torch.cuda.current_stream().synchronize()
        return engine.merge_parts(sigma, k, rank, world, rc, re_, so), parts
    cuts = _piece_cuts(mat_group, order, n)
    dev = logp.device

    def exchange(parts):
        """Starts the transfer of one piece's blocks; returns the receive buffers and the pending work."""
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

    scored, xs = [], []
    for j in range(n):
        pj = engine.score_groups_keymajor(logp[cuts[j]:cuts[j + 1]], mat_group[cuts[j]:cuts[j + 1]], k, log_eps, n_owners=world)
        scored.append(pj)
        xs.append(exchange(pj))                                          # in flight while the next piece is scored
    import time
    t_wait = time.perf_counter()
    for x in xs:
        for w in x["works"]:
            w.wait()
    torch.cuda.current_stream().synchronize()
    scored[0].exchange_exposed_ms = (time.perf_counter() - t_wait) * 1e3     # what the overlap with scoring did not hide
    # sources in global group order: (rank r, piece 0), (rank r, piece 1), ...
    counts = torch.stack([x["rcounts"] for x in xs], dim=1).reshape(n * world, -1).contiguous()
    # all receive buffers are addressed from the lowest base pointer among them (entries are 8 bytes)
    ptrs = [x["rentries"].data_ptr() for x in xs if x["rentries"].numel()]
    base = min(ptrs) if ptrs else xs[0]["rentries"].data_ptr()
    offs = []
    for x in xs:
        if x["rentries"].numel():
            offs.append(np.concatenate([[0], np.cumsum(x["recv"])[:-1]]) + (x["rentries"].data_ptr() - base) // 8)
        else:
            offs.append(np.zeros(world))
    so = np.stack(offs, axis=1).reshape(-1).astype(np.uint64)
    torch.cuda.synchronize()
    db = engine.merge_parts(sigma, k, rank, world, counts, base, so)
    for pj in scored[1:]:
        scored[0].emitted += pj.emitted
        for which in range(7):                                           # IPKGPU_T_* selectors
            scored[0].extra_ms[which] = scored[0].extra_ms.get(which, 0.0) + pj.time_ms(which)
        pj.free()
    return db, scored[0]


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
