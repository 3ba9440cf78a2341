"""Multi-GPU host logic: shard branch groups over ranks, k-mer-keyed exchange of database parts.

The reference processes branch groups sequentially in one thread (ipk/src/db_builder.cpp:602-606; the
OpenMP pragma over groups is commented out).  Groups are independent, so rank r scores a contiguous
range of groups with no data-path collective.  The one real exchange is the k-mer-keyed merge of
the per-rank partial databases (owner = code % world, the kmer_batch rule of
branch_group.cpp:104-107): block o of every rank travels to rank o, after which every rank holds the
complete entry lists of the keys it owns, in global group order (= the reference's append order).

The exchange itself lives in the library (include/ipkgpu.h: ipkgpu_comm_init / ipkgpu_exchange_begin /
ipkgpu_exchange_merge -- grouped ncclSend/ncclRecv over RCCL); this module is the thin caller.  torch.distributed carries
the communicator id to the ranks and serves as the rehearsal transport (gloo: two ranks on one GPU, or CPU tests with
numpy stand-ins for the device steps), where RCCL cannot run.
"""
import os

import numpy as np


def shard_range(n_groups, world, rank):
    """Contiguous, balanced range [g0, g1) of branch groups owned by `rank`."""
    base, extra = divmod(n_groups, world)
    g0 = rank * base + min(rank, extra)
    return g0, g0 + base + (1 if rank < extra else 0)


def piece_cuts(mat_group, n):
    """Matrix cut points of n contiguous ranges of this rank's groups (first-seen order), or None if the matrices of
    two ranges interleave (a piece is scored from a slice of the matrix array).  Fewer groups than pieces: the last
    pieces are empty."""
    mat_group = np.asarray(mat_group)
    order = list(dict.fromkeys(mat_group.tolist()))
    if not order:
        return [0] * (n + 1)
    piece_of = {g: min(n - 1, i * n // len(order)) for i, g in enumerate(order)}
    pid = np.array([piece_of[g] for g in mat_group.tolist()])
    if np.any(np.diff(pid) < 0):
        return None
    return [int(np.searchsorted(pid, j, side="left")) for j in range(n)] + [len(pid)]


def agree_on_pieces(mat_group, want, dist, device):
    """The piece count every rank will use: `want` if every rank can cut its matrices that way, else 1.  Two small
    all-reduces, identical on all ranks -- a rank never issues a collective its peers do not (uneven shards, a rank
    without groups and interleaved matrices all end in the same count everywhere)."""
    import torch
    # the smallest wish wins (ranks with few groups want few pieces), then every rank must be able to cut that many
    t = torch.tensor([max(1, int(want))], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    want = int(t.item())
    ok = 1 if (want <= 1 or piece_cuts(mat_group, want) is not None) else 0
    t = torch.tensor([ok], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return want if int(t.item()) == 1 and want > 1 else 1


def default_pieces(n_groups):
    """Pieces a rank wishes for: a scoring call costs ~0.5 ms before its first group (0.9 ms when this rule was
    set), so small shares are not cut up for the sake of hiding the exchange -- one piece per ~48 groups, at most four (cfg2 on 8
    ranks: 125 groups, 2 pieces: 2 x (0.5 + 62 x 0.02) ms of scoring with half of the exchange under the second piece; three pieces
    would add a third 0.5 ms to hide another sixth of it)."""
    return int(min(4, max(1, n_groups // 48)))


def pieces_model(n_groups, world, entries_per_rank, ms_per_group, call_ms=0.52, link_gb_s=100.0, merge_ms=0.5, max_pieces=3):
    """What cutting a rank's share into p pieces should cost, per step (ms) -- arithmetic on one-GPU measurements, printed into the
    N > 1 bench line so that the first run on a multi-GPU node can be read against it.

    A piece is a scoring call (call_ms before its first group: the intercept of tools/sweep_groups.sh, + ms_per_group each) followed
    by its k-mer-keyed all-to-all, which runs under the NEXT piece's scoring; the last piece's transfer is exposed in full, an earlier
    one only by what outlasts the scoring it hides behind.  A rank sends (world - 1) / world of its entries (8 bytes each) over
    world - 1 direct links at once (link_gb_s each: an ASSUMPTION until measured; xGMI's 153 GB/s is the link's peak)."""
    out = []
    score_all = n_groups * ms_per_group
    send_bytes = 8.0 * entries_per_rank * (world - 1) / max(world, 1)
    xfer_all = send_bytes / max(world - 1, 1) / (link_gb_s * 1e9) * 1e3 if world > 1 else 0.0
    for p in range(1, max_pieces + 1):
        score = p * call_ms + score_all
        xfer = xfer_all / p
        exposed = xfer + (p - 1) * max(0.0, xfer - (score_all / p + call_ms))
        out.append({"pieces": p, "scoring_ms": score, "extra_fixed_ms": (p - 1) * call_ms, "exchange_exposed_ms": exposed,
                    "merge_ms": merge_ms if world > 1 else 0.0, "step_ms": score + exposed + (merge_ms if world > 1 else 0.0)})
    return {"assumptions": {"call_ms": call_ms, "ms_per_group": ms_per_group, "link_GB_s": link_gb_s, "merge_ms": merge_ms,
                            "entries_per_rank": entries_per_rank, "groups_per_rank": n_groups, "world": world}, "by_pieces": out}


def exchange_parts(counts, entries, owner_offsets, dist, world):
    """Rehearsal transport (torch.distributed): all-to-all of the per-owner blocks of ONE piece.

    counts   tensor [world, slots] int32 -- row o goes to rank o
    entries  tensor [n, 2] int32         -- rows owner_offsets[o]:owner_offsets[o+1] go to rank o
    Returns (recv_counts [world, slots], recv_entries [m, 2], recv sizes per source)."""
    import torch
    dev = counts.device
    send_sizes = [int(owner_offsets[o + 1] - owner_offsets[o]) for o in range(world)]
    if world == 1:
        return counts, entries, send_sizes
    if dev.type == "cuda" and dist.get_backend() == "gloo":
        # gloo has no device all-to-all: stage through host memory
        rc, re_, rs = exchange_parts(counts.cpu(), entries.cpu(), owner_offsets, dist, world)
        return rc.to(dev), re_.to(dev), rs
    ss = torch.tensor(send_sizes, dtype=torch.int64, device=dev)
    rs = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rs, ss)
    recv_sizes = [int(x) for x in rs.tolist()]
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts.contiguous())
    recv_entries = torch.empty((sum(recv_sizes), 2), dtype=entries.dtype, device=dev)
    dist.all_to_all_single(recv_entries, entries.contiguous(), output_split_sizes=recv_sizes, input_split_sizes=send_sizes)
    return recv_counts, recv_entries, recv_sizes


def init_native_comm(engine, dist, world, rank):
    """Creates the library's RCCL communicator on every rank (id from rank 0, carried by torch.distributed).
    Returns False -- on ALL ranks alike -- when RCCL is not usable (gloo rehearsal, several ranks on one GPU, no library).

    Failure is kept symmetric: ncclCommInitRank is a collective, so a rank that gave up before it would leave its peers
    waiting inside it.  Everything that can fail on one rank alone (loading RCCL, the exchange stream and buffers:
    comm_prepare) comes first and the ranks agree on its outcome with an all-reduce; only then does rank 0 draw the id and
    every rank enter comm_init.  A failure inside the collective itself raises on the rank that sees it (its process exits
    non-zero, the launcher tears the job down); bench.py's watchdog bounds the wait of the others."""
    import torch
    if getattr(engine, "comm_world", 1) == world and world > 1:
        return True
    if os.environ.get("IPK_DIST_NATIVE", "1") == "0" or dist.get_backend() != "nccl":
        return False
    dev = torch.device("cuda", torch.cuda.current_device())
    ok = 1
    try:
        engine.comm_prepare(world)
    except Exception:
        ok = 0
    t = torch.tensor([ok], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) != 1:
        return False
    buf = torch.zeros(129, dtype=torch.uint8, device=dev)
    if rank == 0:
        try:
            buf[:128] = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).to(dev)
            buf[128] = 1
        except Exception:
            pass
    dist.broadcast(buf, 0)
    if int(buf[128].item()) != 1:
        return False
    engine.comm_init(bytes(buf[:128].cpu().numpy().tobytes()), rank, world)      # collective: raises, never "returns False"
    t = torch.tensor([engine.comm_world_seen()], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) != world:
        raise RuntimeError(f"RCCL communicator saw {int(t.item())} ranks, expected {world}")
    return True


def build_db_shard(engine, logp, mat_group, k, log_eps, sigma, dist=None, world=1, rank=0, overlap=True, pieces=None, agreed=False):
    """Scores this rank's groups and returns (this rank's database shard, parts) -- the state
    `_phylo_kmer_db` has after explore_kmers (db_builder.cpp:576-627), sharded by k-mer owner.

    With several ranks the groups are scored in `pieces` contiguous ranges (default: default_pieces(), or IPK_DIST_PIECES) so that the
    exchange of one range's blocks runs while the next range is being scored -- only the last range's transfer is
    exposed; the merge takes pieces x world sources in the order (rank 0 piece 0, rank 0 piece 1, ..., rank 1 piece 0, ...),
    which is global group order.  All ranks use the same piece count (agree_on_pieces: two small all-reduces per call, unless the
    caller settled the count beforehand -- `pieces` from agree_on_pieces() with agreed=True, as a loop over equally shaped calls does)."""
    if world == 1:
        parts = engine.score_groups_keymajor(logp, mat_group, k, log_eps, n_owners=1)
        return engine.db_from_parts(parts, sigma, k), parts

    import time
    import torch
    mat_group = np.ascontiguousarray(mat_group, dtype=np.uint32)
    on_gpu = dist.get_backend() != "gloo"
    env = os.environ.get("IPK_DIST_PIECES")
    want = pieces if pieces is not None else (int(env) if env else default_pieces(len(np.unique(mat_group))))
    if not (overlap and hasattr(logp, "data_ptr")):
        want = 1
    n = max(1, want) if agreed else agree_on_pieces(mat_group, max(1, want), dist, "cuda" if on_gpu else "cpu")
    cuts = piece_cuts(mat_group, n) if n > 1 else [0, len(mat_group)]
    native = init_native_comm(engine, dist, world, rank)

    scored, xs = [], []
    for j in range(n):
        a, b = cuts[j], cuts[j + 1]
        pj = engine.score_groups_keymajor(logp[a:b], mat_group[a:b], k, log_eps, n_owners=world)
        scored.append(pj)
        if native:
            xs.append(engine.exchange_begin(pj))                            # enqueued on the communicator's stream: runs under the next piece
        else:
            xs.append(exchange_parts(pj.counts_tensor(), pj.entries_tensor(), pj.owner_offsets, dist, world))
    t_wait = time.perf_counter()
    if native:
        db, exposed = engine.exchange_merge(xs, sigma, k)
    else:
        torch.cuda.current_stream().synchronize()
        # sources in global group order: (rank s, piece 0), (rank s, piece 1), ...: one pointer pair per source
        cps, eps = [], []
        for s in range(world):
            for rc, re_, rs in xs:
                cps.append(rc[s].data_ptr())
                off = int(sum(rs[:s]))
                eps.append(re_.data_ptr() + 8 * off)
        db = engine.merge_parts_ptrs(sigma, k, rank, world, cps, eps)
        exposed = (time.perf_counter() - t_wait) * 1e3
    first = scored[0]
    first.exchange_exposed_ms = exposed
    first.exchange = "rccl" if native else "torch"      # in-library grouped ncclSend/ncclRecv | torch.distributed transport
    for pj in scored[1:]:
        first.emitted += pj.emitted
        for which in range(10):                                          # IPKGPU_T_* selectors
            first.extra_ms[which] = first.extra_ms.get(which, 0.0) + pj.time_ms(which)
        pj.free()
    return db, first


def write_db_file(path, sequence_type, tree_index, newick, kmer_size, omega, write_shard, workdir, dist=None, world=1, rank=0):
    """Database file from the ranks' shards.  `write_shard(file)` writes THIS rank's shard -- its k-mers in its own filter
    order -- as a database file (dbfile.write_db_device / dbfile.write_db) and returns nothing.  One rank: that file is the
    database.  Several ranks: every rank drops its shard into workdir/shards (the reference's on-disk mode leaves its batches
    in workdir/hashmaps the same way, db_builder.cpp:460-464) and rank 0 merges the shard files by (filter value, key),
    streaming (dbfile.merge_shard_files = ipkgpu_db_merge_files: merge_stage2's role, db_builder.cpp:392-458).
    Returns (total k-mers, total entries) on the writing rank, None elsewhere."""
    from . import dbfile
    if world == 1:
        write_shard(path)
        return None
    sdir = os.path.join(workdir, "shards")
    os.makedirs(sdir, exist_ok=True)
    mine = os.path.join(sdir, f"shard{rank}.ipk")
    write_shard(mine)
    dist.barrier()
    out = None
    if rank == 0:
        paths = [os.path.join(sdir, f"shard{r}.ipk") for r in range(world)]
        out = dbfile.merge_shard_files(path, sequence_type, tree_index, newick, kmer_size, omega, paths)
    dist.barrier()
    os.remove(mine)
    return out
