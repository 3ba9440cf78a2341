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


def build_db_shard(engine, logp, mat_group, k, log_eps, sigma, dist=None, world=1, rank=0):
    """Scores this rank's groups and returns (this rank's database shard, parts) -- the state
    `_phylo_kmer_db` has after explore_kmers (db_builder.cpp:576-627), sharded by k-mer owner."""
    parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=world)
    if world == 1:
        return engine.db_from_parts(parts, sigma, k), parts
    counts, entries = parts.counts_tensor(), parts.entries_tensor()
    rc, re_, so = exchange_parts(counts, entries, parts.owner_offsets, dist, world)
    if world > 1:
        import torch
        torch.cuda.current_stream().synchronize()
    db = engine.merge_parts(sigma, k, rank, world, rc, re_, so)
    return db, parts
