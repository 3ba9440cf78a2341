#!/usr/bin/env python3
"""bench.py -- scored phylo-k-mers/s of the hot path on N MI355X (one process per GPU).

A step = one pass of the hot path over the whole synthetic workload of this rank, matrices already
resident in HBM: prefix array + DCLA scoring + per-branch max-reduce + assembly of the key-major
phylo-k-mer database (what `_phylo_kmer_db` holds after explore_kmers, db_builder.cpp:576-698); with
N > 1 ranks the database is sharded by k-mer owner and the step includes the all-to-all exchange
(RCCL) and merge.  `--output group` times the group-major form (sorted (key, score) set per branch).  Default workload: BASELINE.json configs[1]
("Synthetic DNA: 2000 extended nodes x 10000 sites, k=10, omega=1.5, 1xMI355X").
Scaling: `--scaling strong` (default; the north star's fixed 2000-node workload): the config's branch groups are split
over the ranks by contiguous ranges (distributed.shard_range), value = the whole workload's scored phylo-k-mers per
second.  `--scaling weak`: every rank scores its own full copy of the config (per-GPU work fixed).  Either way the only
collective is the k-mer-keyed exchange of the database parts.

`e2e` in the JSON line (rank 0, N = 1): the COLD build of the same workload, wall clock -- new context, matrices uploaded
from host memory, first scoring call, MIF0 filter, database file written -- i.e. what a one-shot `ipk.py build` pays
beyond reading its inputs; never part of `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cold_end_to_end(cfg, n_groups, device):
    """Cold build of the benchmarked workload on one GPU, wall clock, in THIS (fresh) process: what `ipk.py build` pays after
    its inputs are parsed.  The reference prints the same stages (Computation / Filtering / Merge time,
    db_builder.cpp:217,230-236,288-290,334-336).  Runs as a child of the benchmark (`--e2e-child`): a context created after
    another one of the same process has freed tens of GB pays a driver-side penalty of seconds on its first large
    allocation, which a one-shot build never sees."""
    import shutil
    import tempfile
    import torch
    import ipk_amd
    from ipk_amd import dbfile
    from ipk_amd.synth import synth_matrices
    sigma, k, mpg, sites = cfg["sigma"], cfg["k"], cfg["mats_per_group"], cfg["sites"]
    eps = ipk_amd.log_threshold(cfg["omega"], sigma, k)
    n_mats = n_groups * mpg
    host = np.empty((n_mats, sites, sigma), dtype=np.float32)     # the matrices as the loader leaves them: pageable host memory
    step_m = max(1, (64 << 20) // (sites * sigma * 4))
    for m0 in range(0, n_mats, step_m):
        m1 = min(n_mats, m0 + step_m)
        host[m0:m1] = synth_matrices(m1 - m0, sites, sigma, cfg["alpha"], cfg["seed"], first_mat=m0)
    groups = np.repeat(np.arange(n_groups, dtype=np.uint32), mpg)
    tmpdir = tempfile.mkdtemp(prefix="ipk_e2e_")
    path = os.path.join(tmpdir, "DB.ipk")
    res = {}
    try:
        t_all = time.perf_counter()
        t0 = time.perf_counter(); torch.cuda.set_device(device); eng = ipk_amd.Engine(device); res["create_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); dev = torch.empty(host.shape, dtype=torch.float32, device="cuda"); torch.cuda.synchronize(); res["alloc_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); dev.copy_(torch.from_numpy(host)); torch.cuda.synchronize(); res["upload_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); parts = eng.score_groups_keymajor(dev, groups, k, eps, n_owners=1); res["score_first_call_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); db = eng.db_from_parts(parts, sigma, k); res["db_s"] = time.perf_counter() - t0
        t0 = time.perf_counter(); db.filter_mif0(eng, n_groups + 1, ipk_amd.score_threshold(cfg["omega"], sigma, k)); res["filter_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        nbytes = dbfile.write_db_device(eng, db, path, "DNA" if sigma == 4 else "AA", [], "", k, cfg["omega"])
        res["write_s"] = time.perf_counter() - t0
        wt = dbfile.write_times(eng)
        res["write_device_s"], res["write_file_s"] = wt["device_s"], wt["file_s"]
        res["cold_s"] = time.perf_counter() - t_all
        res["gpu_part_s"] = res["create_s"] + res["alloc_s"] + res["upload_s"] + res["score_first_call_s"] + res["db_s"] + res["filter_s"] + res["write_device_s"]
        res["file_bytes"] = nbytes
        res["file_on"] = tmpdir
        res["kmers"], res["entries"], res["scored"] = db.num_keys, db.num_entries, parts.emitted
        res["score_device_ms"] = parts.time_ms(0)
        db.free(); parts.free(); eng.close()
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)
    return res


def multi_rank_file_stages(eng, D, dist, d_logp, groups, k, eps, sigma, cfg, world, rank, n_pieces, ng_total):
    """One more build on every rank, then the stages behind it, each bracketed by barriers (max over ranks = wall time): MIF0 on the
    rank's shard, the shard written as a file (records packed on the device), rank 0's streaming merge of the P shard files.  Local
    failures (no room for the files ...) are agreed on with an all-reduce before anything collective depends on them.  Skipped where
    the files would not fit the scratch directory."""
    import shutil
    import tempfile
    import torch
    import ipk_amd
    from ipk_amd import dbfile
    cdev = "cpu" if dist.get_backend() == "gloo" else "cuda"

    def wall(t0):
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    res = {}
    dist.barrier(); t0 = time.perf_counter()
    db, t = D.build_db_shard(eng, d_logp, groups, k, eps, sigma, dist, world, rank, pieces=n_pieces, agreed=n_pieces is not None)
    res["build_s"] = wall(t0)
    t0 = time.perf_counter()
    db.filter_mif0(eng, ng_total + 1, ipk_amd.score_threshold(cfg["omega"], sigma, k))
    res["filter_s"] = wall(t0)
    tot = torch.tensor([db.num_keys, db.num_entries], dtype=torch.int64, device=cdev)
    dist.all_reduce(tot)
    file_bytes = int(tot[0].item()) * 16 + int(tot[1].item()) * 8
    # one scratch directory for all ranks (they share the node): rank 0 makes it
    name = [tempfile.mkdtemp(prefix="ipk_e2e_multi_") if rank == 0 else None]
    dist.broadcast_object_list(name, 0)
    tmpdir = name[0]
    ok = 1
    try:
        free = shutil.disk_usage(tmpdir).free
        if 2.2 * file_bytes > free:
            ok = 0
            res["skipped"] = f"shards + merged file need {2.2 * file_bytes / 1e9:.0f} GB, {free / 1e9:.0f} GB free in {tmpdir}"
    except Exception as exc:
        ok = 0; res["error"] = repr(exc)
    okt = torch.tensor([ok], dtype=torch.int64, device=cdev); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) == 1:
        mine = os.path.join(tmpdir, f"shard{rank}.ipk")
        dist.barrier(); t0 = time.perf_counter()
        try:
            dbfile.write_db_device(eng, db, mine, "DNA" if sigma == 4 else "AA", [], "", k, cfg["omega"])
        except Exception as exc:
            ok = 0; res["error"] = repr(exc)
        res["shard_files_s"] = wall(t0)
        okt = torch.tensor([ok], dtype=torch.int64, device=cdev); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 1:
            dist.barrier(); t0 = time.perf_counter()
            if rank == 0:
                try:
                    nk, ne = dbfile.merge_shard_files(os.path.join(tmpdir, "DB.ipk"), "DNA" if sigma == 4 else "AA", [], "", k, cfg["omega"],
                                                      [os.path.join(tmpdir, f"shard{r}.ipk") for r in range(world)])
                    res["kmers"], res["entries"] = nk, ne
                    res["file_bytes"] = os.path.getsize(os.path.join(tmpdir, "DB.ipk"))
                except Exception as exc:
                    res["error"] = repr(exc)
            res["merge_rank0_s"] = wall(t0)
    dist.barrier()
    if rank == 0:
        shutil.rmtree(tmpdir, ignore_errors=True)
    db.free(); t.free()
    res["note"] = "after the timed steps (workspaces warm): one build, then filter / shard files / rank 0's merge; wall = max over ranks"
    return res if rank == 0 else None


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N` without a launcher: this process touches no GPU and starts the N ranks itself -- as a CHILD
    (`python -m torch.distributed.run`, rendezvous on 127.0.0.1), never by exec -- relays their output (rank 0's JSON line) and
    returns their exit code.  The children carry their own watchdog; the parent's timeout is the last resort."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("IPK_BENCH_WATCHDOG_S", "900")) + 60.0
    pr = subprocess.Popen(cmd, env=env, start_new_session=True)       # stdout/stderr inherited: the JSON line passes straight through
    try:
        return pr.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        print(f"bench.py: ranks still running after {limit:.0f} s -- killing the job", file=sys.stderr)
        try:
            os.killpg(pr.pid, signal.SIGKILL)                          # the session this parent created, nothing else
        except ProcessLookupError:
            pass
        pr.wait()
        return 124
    except KeyboardInterrupt:
        import signal
        os.killpg(pr.pid, signal.SIGTERM)
        return 130


def start_watchdog():
    """A rank that waits for ever (a peer died inside a collective, a wedged transfer) must end with a non-zero code instead
    of holding the node: a daemon thread ends the process after IPK_BENCH_WATCHDOG_S seconds (default 900; 0 = off)."""
    import threading
    limit = float(os.environ.get("IPK_BENCH_WATCHDOG_S", "900"))
    if limit <= 0:
        return

    def bark():
        print(f"bench.py: watchdog -- rank {os.environ.get('RANK', '0')} not finished after {limit:.0f} s, exiting", file=sys.stderr, flush=True)
        os._exit(124)
    t = threading.Timer(limit, bark)
    t.daemon = True
    t.start()


def dry_run(args):
    """Launcher and watchdog rehearsal (CPU, gloo): the ranks meet, count themselves and rank 0 prints a stub line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        if os.environ.get("IPK_BENCH_DRY_HANG") == str(rank):
            time.sleep(3600)                                   # a rank that never reaches the collective
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": seen}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--groups", type=int, default=0, help="override the number of branch groups (0 = config's)")
    ap.add_argument("--alpha", type=float, default=0.0, help="override the column concentration")
    ap.add_argument("--output", default="db", choices=["db", "group"], help="db: key-major database shard; group: per-branch CSR")
    ap.add_argument("--cpu-groups", type=int, default=-1, help="groups timed on the CPU oracle (-1 = auto, 0 = skip)")
    ap.add_argument("--variant", type=int, default=0, help="engine option 'variant' (0 = auto; diagnostics: 1 atomics, 2 chunked pool, 3 exact partition)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="strong: the config's groups split over the ranks; weak: a full copy per rank")
    ap.add_argument("--e2e", type=int, default=1, help="1: also time the cold end-to-end build (N = 1 only; in a fresh child process); 0: skip")
    ap.add_argument("--e2e-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)   # launcher / watchdog rehearsal without a GPU (tests)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.e2e_child:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    start_watchdog()
    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist
    import ipk_amd
    from ipk_amd import engine as E
    from ipk_amd.synth import CONFIGS, synth_matrices

    if args.e2e_child:
        cfg = dict(CONFIGS[args.config])
        if args.alpha:
            cfg["alpha"] = args.alpha
        print("E2E " + json.dumps(cold_end_to_end(cfg, args.groups or cfg["n_groups"], int(os.environ.get("IPK_BENCH_DEVICE", "0")))))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # rehearsal aids for a one-GPU box (never set by the driver): IPK_BENCH_DEVICE pins every rank to one GPU,
    # IPK_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("IPK_BENCH_DEVICE"):
        local_rank = int(os.environ["IPK_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        backend = os.environ.get("IPK_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = dict(CONFIGS[args.config])
    if args.groups:
        cfg["n_groups"] = args.groups
    if args.alpha:
        cfg["alpha"] = args.alpha
    ng, mpg, sites, sigma, k = cfg["n_groups"], cfg["mats_per_group"], cfg["sites"], cfg["sigma"], cfg["k"]
    eps = ipk_amd.log_threshold(cfg["omega"], sigma, k)
    from ipk_amd import distributed as D
    ng_total = ng if args.scaling == "strong" else ng * world
    if args.scaling == "strong":
        g_lo, g_hi = D.shard_range(ng, world, rank)           # this rank's contiguous range of the config's groups
    else:
        g_lo, g_hi = rank * ng, (rank + 1) * ng               # its own copy: groups [rank * ng, (rank + 1) * ng) of an N-fold workload
    ng = g_hi - g_lo
    n_mats = ng * mpg
    first_mat_of_rank = g_lo * mpg

    # synthetic matrices of this rank
    t0 = time.time()
    d_logp = torch.empty((n_mats, sites, sigma), dtype=torch.float32, device="cuda")
    step_m = max(1, min(n_mats, (64 << 20) // (sites * sigma * 4)))
    for m0 in range(0, n_mats, step_m):
        m1 = min(n_mats, m0 + step_m)
        d_logp[m0:m1].copy_(torch.from_numpy(synth_matrices(m1 - m0, sites, sigma, cfg["alpha"], cfg["seed"],
                                                            first_mat=first_mat_of_rank + m0)))
    groups = np.repeat(np.arange(g_lo, g_hi, dtype=np.uint32), mpg)
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    eng = ipk_amd.Engine(local_rank)
    if args.variant:
        eng.set_option("variant", args.variant)
    if os.environ.get("IPKGPU_VARIANT"):
        eng.set_option("variant", int(os.environ["IPKGPU_VARIANT"]))   # diagnostics only
    for knob in ("wg_chunks2", "rounds", "kmc_pass", "prefix_mats"):   # tuning experiments only
        if os.environ.get("IPKGPU_" + knob.upper()):
            eng.set_option("debug_" + knob, int(os.environ["IPKGPU_" + knob.upper()]))
    if os.environ.get("IPKGPU_DEBUG_FLAGS"):
        eng.set_option("debug_flags", int(os.environ["IPKGPU_DEBUG_FLAGS"]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc = {"score": 0.0, "launches": 0.0, "total": 0.0, "compact": 0.0, "prefix": 0.0, "merge": 0.0, "main": 0.0, "reduce": 0.0, "exchange_exposed": 0.0,
           "xp_count": 0.0, "xp_write": 0.0, "km_write": 0.0}
    emitted = entries = n_keys = 0
    exchange_kind = "none"

    # several ranks: the piece count of the exchange overlap is settled once (every step has the same shape), not by two
    # all-reduces per step
    n_pieces = None
    if world > 1 and args.output != "group":
        env = os.environ.get("IPK_DIST_PIECES")
        want = int(env) if env else D.default_pieces(len(np.unique(groups)))
        n_pieces = D.agree_on_pieces(groups, max(1, want), dist, "cpu" if dist.get_backend() == "gloo" else "cuda")

    def step(record):
        nonlocal emitted, entries, n_keys, exchange_kind
        if args.output == "group":
            r = eng.score_groups(d_logp, groups, k, eps)      # returns after the device work completed
            emitted, entries = r.emitted, r.num_entries
            t = r
        else:
            db, t = D.build_db_shard(eng, d_logp, groups, k, eps, sigma, dist if world > 1 else None, world, rank,
                                     pieces=n_pieces, agreed=n_pieces is not None)
            emitted, entries, n_keys = t.emitted, db.num_entries, db.num_keys
            exchange_kind = getattr(t, "exchange", "none")
            if record:
                acc["merge"] += db.time_ms()
                acc["exchange_exposed"] += getattr(t, "exchange_exposed_ms", 0.0)
            db.free()
        if record:
            acc["score"] += t.time_ms(E.T_SCORE); acc["launches"] += t.time_ms(E.T_SCORE_LAUNCHES)
            acc["total"] += t.time_ms(E.T_TOTAL); acc["compact"] += t.time_ms(E.T_COMPACT); acc["prefix"] += t.time_ms(E.T_PREFIX)
            acc["main"] += t.time_ms(E.T_SCORE_MAIN); acc["reduce"] += t.time_ms(E.T_SCORE_REDUCE)
            if args.output != "group":
                acc["xp_count"] += t.time_ms(E.T_XP_COUNT); acc["xp_write"] += t.time_ms(E.T_XP_WRITE); acc["km_write"] += t.time_ms(E.T_KM_WRITE)
        t.free()

    # engine initialisation (untimed, like data generation): the first call of a context allocates and calibrates its
    # workspaces -- hipMalloc of the multi-GB pair pool sporadically takes seconds
    t0 = time.time()
    step(False)
    t_init = time.time() - t0
    if n_pieces is not None and not os.environ.get("IPK_DIST_PIECES"):
        # The piece rule knows group counts, not times (one piece per ~48 groups).  A second, timed step of this very workload says
        # what a rank's scoring costs: one piece per ~3 ms of it, at most four -- pieces_model's optimum for a transfer of about the
        # scoring's length (cfg3: 12.7 ms per rank -> 4 pieces; cfg2: 2.9 ms -> 1).  The smallest wish of all ranks wins, as before.
        acc_probe = dict(acc)
        step(True)
        mine = (acc["total"] - acc_probe["total"]) - (n_pieces or 1) * 0.52
        for kk in acc:
            acc[kk] = acc_probe[kk]
        want = int(min(4, max(1, round(mine / 3.0))))
        n_pieces = D.agree_on_pieces(groups, want, dist, "cpu" if dist.get_backend() == "gloo" else "cuda")
    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    last_kernel = eng.last_main_kernel()
    tables_compressed = eng.last_tables_compressed()
    score_ms, launches, total_ms, compact_ms, prefix_ms = acc["score"], acc["launches"], acc["total"], acc["compact"], acc["prefix"]
    elapsed = time.perf_counter() - t_start
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        cdev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        tl = torch.zeros(world, dtype=torch.float64, device=cdev)
        tl[rank] = acc["total"] / args.steps                       # device time of this rank's scoring calls per step
        dist.all_reduce(tl)
        rank_ms = [float(x) for x in tl.tolist()]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = torch.tensor([emitted], dtype=torch.int64, device=cdev)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        emitted_all = int(e.item())
        # how many ranks the transport itself was set up with: the library's RCCL communicator, or torch's process group
        n_ranks_seen = eng.comm_world_seen() if exchange_kind == "rccl" else dist.get_world_size()
    else:
        emitted_all = emitted
        n_ranks_seen = 1

    # Several ranks, after the timed region: what the rest of a build costs on top of a step -- filter values on every rank's shard,
    # the shards written as files, rank 0's merge of the shard files (merge_stage2's role, db_builder.cpp:392-458).  Warm, not cold.
    e2e_multi = None
    if world > 1 and args.output == "db" and args.e2e:
        e2e_multi = multi_rank_file_stages(eng, D, dist, d_logp, groups, k, eps, sigma, cfg, world, rank, n_pieces, ng_total)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = emitted_all * args.steps / elapsed
        # roofline of the dominant kernel (scoring + max-reduce): algorithmic bytes per launch =
        # every matrix read once + one (u32 key, f32 score) pair per scored phylo-k-mer
        mats_bytes = n_mats * sites * sigma * 4
        b_alg = mats_bytes + 8 * emitted
        main_kernel = last_kernel or ("score_stream_kernel" if acc["reduce"] > 0 else "score_tiles_kernel")
        avg_score_ms = score_ms / max(launches, 1)
        avg_main_ms = acc["main"] / max(launches, 1)          # the dominant kernel alone (HIP events on its stream)
        # a step is one launch per batch or, with several ranks, per piece of the rank's groups: bytes per LAUNCH
        b_alg = b_alg * args.steps / max(launches, 1)
        achieved = b_alg / (avg_main_ms * 1e-3) / 1e9 if avg_main_ms > 0 else 0.0
        traffic, traffic_kernels = None, {}
        # written by tools/pmc_traffic.py from rocprofv3 --pmc passes (cfg2: pmc_traffic.json, others: pmc_traffic_<config>.json)
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json" if args.config == "cfg2" else f"pmc_traffic_{args.config}.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("workload") == args.config and not args.groups and not args.alpha and not args.variant and world == 1:
                    traffic_kernels = {kname: kv.get("hbm_bytes_per_launch") for kname, kv in tj.get("kernels", {}).items()}
                    traffic = traffic_kernels.get(main_kernel)
            except Exception:
                traffic = None
        # per-kernel lines: every kernel with its own algorithmic bytes (no lumping); ms = average launch duration from the
        # library's HIP events.  pairs = scored phylo-k-mers (8 B each), entries = (branch, k-mer) entries (8 B each).
        per_launch = lambda key: acc[key] / max(launches, 1)
        e_bytes, p_bytes = 8.0 * entries * args.steps / max(launches, 1), 8.0 * emitted * args.steps / max(launches, 1)
        kernels = []
        def kline(name, ms, bytes_):
            if ms > 0:
                g = bytes_ / (ms * 1e-3) / 1e9
                kernels.append({"kernel": name, "avg_launch_ms": ms, "algorithmic_bytes_per_launch": bytes_, "achieved": g, "unit": "GB/s",
                                "frac": g / HBM_PEAK_GBPS, "traffic": traffic_kernels.get(name)})
        mb = mats_bytes * args.steps / max(launches, 1)
        if main_kernel == "score_xp_kernel":
            main_kernel = "score_xp_kernel<WRITE>"                          # the dominant launch of the exact-partition variant
            avg_main_ms = per_launch("xp_write")
            achieved = b_alg / (avg_main_ms * 1e-3) / 1e9 if avg_main_ms > 0 else 0.0
            kline("score_xp_kernel<COUNT>", per_launch("xp_count"), mb)                 # reads the matrices, writes counters only
            kline("score_xp_kernel<WRITE>", per_launch("xp_write"), mb + p_bytes)       # the pairs leave here
            kline("reduce_ranges_kernel", per_launch("reduce"), p_bytes + e_bytes / 2)  # pairs in, one 4-B score code per entry out
            kline("km_write_c_kernel", per_launch("km_write"), e_bytes / 2 + e_bytes)   # score codes in, 8-B entries out
        elif tables_compressed:
            kline(main_kernel, avg_main_ms, b_alg)
            kline("reduce_buckets_kernel<COMPRESS>", per_launch("reduce"), p_bytes + e_bytes / 2)        # pairs in, one 4-B score code per entry out
            kline("km_write_c_kernel", per_launch("km_write"), e_bytes / 2 + e_bytes)                  # score codes in, 8-B entries out
        else:
            kline(main_kernel, avg_main_ms, b_alg)
            kline("reduce_buckets_kernel", per_launch("reduce"), p_bytes + 4.0 * (sigma ** k) * ng)   # pairs in, dense tables out
            dbg = int(os.environ.get("IPKGPU_DEBUG_FLAGS", "0"))
            dense_writer = "km_write_kernel" if (dbg & 512) or (ng < 96 and not dbg & 1024) else "km_write_lines_kernel"
            kline(dense_writer, per_launch("km_write"), 4.0 * (sigma ** k) * ng + e_bytes)             # dense tables in, entries out
        out = {
            "metric": "scored phylo-k-mers/sec", "value": value, "unit": "phylo-k-mers/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "exchange": exchange_kind, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic {'DNA' if sigma == 4 else 'AA'} {n_mats} extended nodes "
                                   f"({ng} branch groups x {mpg}) x {sites} sites, k={k}, omega={cfg['omega']}, "
                                   f"alpha={cfg['alpha']}, per GPU" + (f"; whole workload {ng_total} branch groups" if world > 1 else ""),
                       "scored_per_step_per_gpu": emitted, "branch_kmer_entries_per_gpu": entries,
                       "output": "key-major database shard" if args.output == "db" else "group-major CSR",
                       "sharding": f"branch groups over {world} rank(s)" + ("; k-mer-keyed all-to-all (RCCL) + merge" if world > 1 and args.output == "db" else "; no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # (not measured in this run: PMC counters need their own rocprofv3 passes -- the figure is the committed capture)
                         "traffic_source": (os.path.relpath(tfile, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_traffic.sh, "
                                            "not collected in this run)") if traffic is not None else None,
                         "kernel": main_kernel, "avg_launch_ms": avg_main_ms, "kernels": kernels,
                         "algorithmic_bytes_per_launch": b_alg,
                         "score_phase_ms": avg_score_ms, "score_phase_GBps": b_alg / (avg_score_ms * 1e-3) / 1e9 if avg_score_ms > 0 else 0.0},
            "phases_ms_per_step": {"prefix": prefix_ms / args.steps, "score": score_ms / args.steps,
                                   "score_main_kernel": acc["main"] / args.steps, "score_lds_reduce": acc["reduce"] / args.steps,
                                   "compact": compact_ms / args.steps, "device_total": total_ms / args.steps,
                                   "db_merge": acc["merge"] / args.steps,
                                   "exchange_exposed": acc["exchange_exposed"] / args.steps,
                                   "per_rank_device_total": rank_ms},
            "setup_s": {"synth_and_upload": t_gen, "engine_init_first_call": t_init},
        }
        if world > 1 and args.output == "db":
            # the piece rule's arithmetic beside what was measured (pieces used: n_pieces): per-group cost from this run's device time
            per_group = max(0.0, (total_ms / args.steps - (n_pieces or 1) * 0.52) / max(ng, 1))
            if e2e_multi is not None:
                out["e2e_multi"] = e2e_multi
            out["pieces_used"] = n_pieces or 1
            out["pieces_model"] = D.pieces_model(ng, world, entries, per_group)
        if world == 1 and args.e2e:
            import subprocess
            try:
                eng.close()                                       # the child gets the GPU's memory to itself
                del d_logp
                torch.cuda.empty_cache()
                cmd = [sys.executable, os.path.abspath(__file__), "--e2e-child", "--config", args.config, "--groups", str(ng)]
                if args.alpha:
                    cmd += ["--alpha", str(args.alpha)]
                env = dict(os.environ, IPK_BENCH_DEVICE=str(local_rank))
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
                line = [l for l in pr.stdout.splitlines() if l.startswith("E2E ")]
                out["e2e"] = json.loads(line[0][4:]) if line else {"error": (pr.stderr or pr.stdout)[-800:]}
            except Exception as exc:                              # e.g. no room for the file: the steady-state line stays valid
                out["e2e"] = {"error": repr(exc)}
            # the loader (raxmlng_reader, ar.cpp:144-270; inside the reference's stage-1 timer, outside `value` here -- SURVEY 8d:
            # "reported separately"): a bounded sample of the config's .raxml.ancestralProbs, text -> log10 float32 matrices
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import loader_bench
                n_sample = min(n_mats, 200)
                lb = loader_bench.measure(n_sample, sites, sigma)
                full_bytes = lb["file_bytes"] / n_sample * n_mats
                allc = lb["read"][-1]
                lb["sample"] = f"{n_sample} of {n_mats} nodes ({lb['file_bytes'] / 1e6:.0f} MB of text); full file {full_bytes / 1e9:.2f} GB"
                lb["loader_s_est"] = full_bytes / (lb["open_index_GBps"] * 1e9) + full_bytes / (allc["text_GBps"] * 1e9)
                lb["loader_s_est_note"] = f"index + read of the full file at the sample's rates, {allc['threads']} threads"
                if isinstance(out.get("e2e"), dict):
                    out["e2e"]["loader"] = lb
                    out["e2e"]["loader_s"] = lb["loader_s_est"]
            except Exception as exc:
                if isinstance(out.get("e2e"), dict):
                    out["e2e"]["loader"] = {"error": repr(exc)}
        # CPU baseline: the oracle (C restatement), 1 thread, bounded sample of the same workload
        n_cpu = args.cpu_groups
        if world == 1 and n_cpu != 0:
            from oracle import ipk_oracle as co
            if n_cpu < 0:
                n_cpu = 1
            probe = synth_matrices(mpg, sites, sigma, cfg["alpha"], cfg["seed"], first_mat=0)
            tp = time.perf_counter(); co.explore_many(probe, mpg, k, eps); tp = time.perf_counter() - tp
            if args.cpu_groups < 0:
                n_cpu = int(max(2, min(ng, 15.0 / max(tp, 1e-3))))
            sample = synth_matrices(n_cpu * mpg, sites, sigma, cfg["alpha"], cfg["seed"], first_mat=0)
            tc = time.perf_counter(); e_cpu, u_cpu = co.explore_many(sample, mpg, k, eps); tc = time.perf_counter() - tc
            # the same sample through the GPU path: scored count and unique (branch, k-mer) entries must agree
            if not eng._h:
                eng = ipk_amd.Engine(local_rank)                 # (the cold end-to-end leg closed the benchmark's context)
            rs = eng.score_groups(sample, np.repeat(np.arange(n_cpu, dtype=np.uint32), mpg), k, eps)
            out["sample_check"] = {"scored_equal": rs.emitted == e_cpu, "entries_equal": rs.num_entries == u_cpu,
                                   "scored": e_cpu, "entries": u_cpu}
            rs.free()
            out["cpu_baseline"] = {"value": e_cpu / tc, "unit": "phylo-k-mers/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_cpu} of {ng} branch groups of the same workload "
                                             f"({e_cpu} scored k-mers, {tc:.1f} s), oracle/ipk_oracle.c -O3, 1 thread "
                                             f"(the reference build loop is single-threaded)"}
            # the generous baseline: the same sample with the groups dealt to every host core this process may use
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if cores > 1 and n_cpu >= 2:
                from concurrent.futures import ThreadPoolExecutor
                nt = min(cores, n_cpu, int(os.environ.get("IPK_BENCH_CPU_THREADS", "16")))      # a one-GPU box has a 16-core share
                cuts = [n_cpu * i // nt for i in range(nt + 1)]
                tm = time.perf_counter()
                with ThreadPoolExecutor(nt) as ex:              # ctypes releases the GIL inside the C oracle
                    parts = list(ex.map(lambda i: co.explore_many(sample[cuts[i] * mpg:cuts[i + 1] * mpg], mpg, k, eps), range(nt)))
                tm = time.perf_counter() - tm
                assert sum(p_[0] for p_ in parts) == e_cpu
                out["cpu_baseline"]["all_cores"] = {"value": e_cpu / tm, "unit": "phylo-k-mers/s", "cores": nt,
                                                    "sample": f"the same {n_cpu} groups, one thread per group range, {tm:.1f} s"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
