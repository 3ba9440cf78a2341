"""`ipk build`-compatible command line for the MI355X engine (SURVEY.md section 8f, rows n3 + n4).

Option names and defaults follow the reference wrapper (ipk.py:70-230 -> argv table ipk.py:292-329,
ipk/src/command_line.cpp:83-147), and so does the flow of main.cpp:129-200 for the stages this repository owns:

  reference tree (-t) -> ghost nodes (extended_tree.cpp:76-162; saved as workdir/extended_trees/extended_tree.newick)
  -> AR outputs found by suffix in --ar-dir (ar.cpp:611-640: *.raxml.ancestralProbs, *.raxml.ancestralTree)
  -> AR tree rerooted if the reference tree is rooted (main.cpp:172-178) -> extended/AR node mapping (ar.cpp:790-834)
  -> ghost groups (db_builder.cpp:495-553) -> GPU scoring -> MIF0 filter -> database file.

Alignment reduction/extension and RUNNING the ancestral reconstruction are IPK's host stages and out of scope
(DESIGN.md): their options are accepted for command-line compatibility and ignored; the AR outputs must exist (e.g. from
`ipk.py build --ar-only` of the reference, or raxml-ng run on the saved extended tree).

  ipk.py build -r aln.fasta -t tree.nwk -w work --ar-dir work/AR -k 10

--mapping (optional, not in the reference): TSV `ar_node_label <TAB> branch_postorder_id` per ghost node in scoring order;
replaces the tree-derived plan (synthetic inputs without trees).
"""
import glob
import os
import sys
import time

import click
import numpy as np


@click.group()
def ipk():
    """MI355X phylo-k-mer database builder (drop-in for the scoring path of IPK)."""


@ipk.command()
@click.option("-b", "--ar", type=click.Path(), help="(ignored) ancestral-reconstruction binary; use --ar-dir")
@click.option("-r", "--refalign", type=click.Path(), help="(ignored here) reference alignment")
@click.option("-t", "--reftree", type=click.Path(), help="reference tree (newick); stored in the database header")
@click.option("-s", "--states", type=click.Choice(["nucl", "amino"]), default="nucl", show_default=True)
@click.option("-v", "--verbosity", type=int, default=1, show_default=True)
@click.option("-w", "--workdir", required=True, type=click.Path(file_okay=False))
@click.option("--write-reduction", type=click.Path(), help="(ignored)")
@click.option("-a", "--alpha", type=float, default=1.0, show_default=True, help="(ignored) AR gamma shape")
@click.option("-c", "--categories", type=int, default=4, show_default=True, help="(ignored) AR rate categories")
@click.option("-k", "--k", "k", type=int, default=8, show_default=True, help="k-mer length (DNA <= 14, AA <= 6 on this engine)")
@click.option("-m", "--model", default=None, help="(ignored) AR model")
@click.option("--convert-uo", is_flag=True, help="(ignored)")
@click.option("--no-reduction", is_flag=True, help="(ignored)")
@click.option("--reduction-ratio", type=float, default=0.99, show_default=True, help="(ignored)")
@click.option("--omega", type=float, default=1.5, show_default=True, help="score threshold (omega/#states)^k")
@click.option("--filter", "filter_", type=click.Choice(["mif0", "random"]), default="mif0", show_default=True)
@click.option("-u", "--mu", type=float, default=1.0, show_default=True, help="(parsed, unused -- as in the reference build)")
@click.option("--ghosts", type=click.Choice(["inner-only", "outer-only", "both"]), default="both", show_default=True)
@click.option("--use-unrooted", is_flag=True, help="(ignored)")
@click.option("--merge-branches", is_flag=True, help="unsupported (as in the reference, main.cpp:31-37)")
@click.option("--ar-dir", type=click.Path(exists=True, file_okay=False), default=None,
              help="directory holding <prefix>.raxml.ancestralProbs / .raxml.ancestralTree [searched in the workdir if absent]")
@click.option("--ar-only", is_flag=True, help="(ignored)")
@click.option("--ar-config", type=click.Path(), help="(ignored)")
@click.option("--keep-positions", is_flag=True,
              help="(ipk-aa-pos; amino acids, one GPU) every database entry carries the window position of its kept score "
                   "(db_builder.cpp:655-662,687-689): positions are scored per branch (ipkgpu_score_groups_positions) and joined with "
                   "the key-major database on the host; the entry layout (branch, score, u16 position) is a guess like the rest of the file")
@click.option("--uncompressed", is_flag=True, help="(ignored, as in the reference)")
@click.option("--threads", type=int, default=0,
              help="host threads of the probability loader [0 = every core this process may run on, divided among the ranks of a node; the reference's --threads only "
                   "feeds the AR tool (ar.cpp:669,749), which this command does not run]")
@click.option("-o", "--output", default=None, help="output file [workdir/DB.ipk]")
@click.option("--on-disk", is_flag=True, help="(ignored: the GPU build batches groups by HBM instead)")
@click.option("--mapping", type=click.Path(exists=True), default=None,
              help="TSV: AR node label <TAB> branch post-order id, one line per ghost node (instead of the tree-derived plan)")
@click.option("--num-tree-nodes", type=int, default=0, help="node count of the original tree (MIF0's N, db_builder.cpp:261); default: the reference tree's, or branch groups + 1 with --mapping")
@click.option("--device", type=int, default=None, help="GPU index [0; LOCAL_RANK under torchrun]")
def build(ar, refalign, reftree, states, verbosity, workdir, write_reduction, alpha, categories, k, model, convert_uo,
          no_reduction, reduction_ratio, omega, filter_, mu, ghosts, use_unrooted, merge_branches, ar_dir, ar_only,
          ar_config, keep_positions, uncompressed, threads, output, on_disk, mapping, num_tree_nodes, device):
    """Computes a database of phylo-k-mers from precomputed ancestral probabilities."""
    import ipk_amd
    from ipk_amd import dbfile, distributed
    from ipk_amd.loader import AncestralProbs

    if keep_positions and states == "nucl":
        raise click.UsageError("--keep-positions is not supported for DNA.")              # ipk.py:281-282
    if keep_positions and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise click.UsageError("--keep-positions: the positioned database is assembled by ONE process (positions are scored per branch, "
                               "ipkgpu_score_groups_positions, and joined with the key-major database on the host); run without torchrun")
    if merge_branches:
        raise click.UsageError("--merge-branches is not supported (the reference only guards it, main.cpp:31-37)")
    from ipk_amd import tree as T
    sigma = 4 if states == "nucl" else 20
    if not 2 <= k <= ipk_amd.max_k(sigma):
        raise click.UsageError(f"k must be in [2, {ipk_amd.max_k(sigma)}] for --states {states} on this engine")
    os.makedirs(workdir, exist_ok=True)
    output = output or os.path.join(workdir, "DB.ipk")

    def find_by_suffix(suffix):                                                       # ar.cpp:458-469, :611-640
        dirs = [ar_dir] if ar_dir else [os.path.join(workdir, "extended_trees"), os.path.join(workdir, "AR"), workdir]
        for d in dirs:
            hits = sorted(glob.glob(os.path.join(d, "*" + suffix)))
            if hits:
                return hits[0]
        raise click.UsageError(f"Could not find \"*{suffix}\" in {dirs}: this build does not run the ancestral reconstruction itself")

    probs_file = find_by_suffix(".raxml.ancestralProbs")
    orig = ext = None
    tree_index, newick, n_tree_nodes = [], "", 0
    if reftree:
        orig = T.Tree.load(reftree)
        if not orig.is_rooted and not use_unrooted:                                   # extended_tree.cpp:169-177
            raise click.UsageError("This reference tree is not rooted. Please provide a rooted tree or provide --use-unrooted. "
                                   "WARNING! This may impact placement accuracy.")
        tree_index, newick, n_tree_nodes = orig.index(), orig.newick(), orig.num_nodes
    labels, branches = [], []
    if mapping:
        for line in open(mapping):
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            lab, br = line.split("\t")[:2]
            if ghosts == "inner-only" and not lab.endswith("_X0"):
                continue
            if ghosts == "outer-only" and not lab.endswith("_X1"):
                continue
            labels.append(lab); branches.append(int(br))
    else:
        if orig is None:
            raise click.UsageError("-t/--reftree is required (or --mapping for inputs without trees)")
        ext = orig.extend()
        ext_dir = os.path.join(workdir, "extended_trees")                               # main.cpp:39-46
        os.makedirs(ext_dir, exist_ok=True)
        if int(os.environ.get("RANK", "0")) == 0:
            with open(os.path.join(ext_dir, "extended_tree.newick"), "w") as fh:
                fh.write(ext.newick() + "\n")
        ar_tree = T.Tree.load(find_by_suffix(".raxml.ancestralTree"))
        if orig.is_rooted and not ar_tree.is_rooted:                                   # main.cpp:172-178
            ar_tree.reroot()
        for _ext_label, ar_label, branch in T.ghost_plan(orig, ext, ar_tree, ghosts):
            labels.append(ar_label); branches.append(branch)
    if not labels:
        raise click.UsageError("no ghost nodes selected")

    # several GPUs: one process per GPU (torchrun); branch groups are split into contiguous ranges of the group
    # order, every rank scores its range, the k-mer-keyed exchange (RCCL) leaves rank r with the k-mers code % P == r
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dist = None
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device)
        own_group = not dist.is_initialized()
        if own_group:
            dist.init_process_group(os.environ.get("IPK_DIST_BACKEND", "nccl"))
    all_branches = list(branches)
    group_order = list(dict.fromkeys(all_branches))                        # first-seen order (db_builder.cpp:524-553)
    g0, g1 = distributed.shard_range(len(group_order), world, rank)
    mine = set(group_order[g0:g1])
    sel = [i for i, b in enumerate(all_branches) if b in mine]
    labels, branches = [labels[i] for i in sel], [all_branches[i] for i in sel]

    t0 = time.time()
    arp = AncestralProbs(probs_file, sigma)
    mats = arp.read(labels, n_threads=threads if threads > 0 else max(1, len(os.sched_getaffinity(0)) // world)) if labels else np.zeros((0, arp.sites, sigma), np.float32)
    t_load = time.time() - t0
    log_eps = ipk_amd.log_threshold(omega, sigma, k)
    eng = ipk_amd.Engine(device)
    t0 = time.time()
    if world > 1:
        import torch
        mats = torch.from_numpy(np.ascontiguousarray(mats)).cuda()
    db, parts = distributed.build_db_shard(eng, mats, np.array(branches, dtype=np.uint32), k, log_eps, sigma, dist, world, rank)
    t_score = time.time() - t0
    # MIF0's N = _original_tree.get_node_count() (db_builder.cpp:261); without a tree: groups = the non-root nodes (:524-553)
    n_nodes = num_tree_nodes or n_tree_nodes or len(group_order) + 1
    t0 = time.time()
    seq_name = "DNA" if sigma == 4 else "AA"
    if filter_ == "mif0":
        db.filter_mif0(eng, n_nodes, ipk_amd.score_threshold(omega, sigma, k))
    t_filter = time.time() - t0
    t0 = time.time()
    if keep_positions:
        # KEEP_POSITIONS (branch_group.cpp:73-86): the kept score's window position.  The database above has the same scores (the position
        # only rides along with the max); the per-branch positioned result is joined to it entry by entry on the host.
        res = eng.score_groups_positions(mats.cpu().numpy() if hasattr(mats, "cpu") else mats, np.array(branches, dtype=np.uint32), k, log_eps)
        keys_db, off_db = db.keys(), db.key_offsets().astype(np.int64)
        br_db, sc_db = db.entries()
        entry_key = np.repeat(keys_db, np.diff(off_db))
        pos_db = np.empty(len(br_db), dtype=np.uint32)
        rk, rs, rp = res.keys(), res.scores(), res.positions()
        for gi, gid in enumerate(res.group_ids.tolist()):
            a, b = int(res.offsets[gi]), int(res.offsets[gi + 1])
            sel = np.flatnonzero(br_db == gid)
            idx = np.searchsorted(rk[a:b], entry_key[sel])
            if len(sel) != b - a or not np.array_equal(rk[a:b][idx], entry_key[sel]) or \
               not np.array_equal(rs[a:b][idx].view(np.uint32), sc_db[sel].view(np.uint32)):
                raise click.ClickException("positioned scoring and the database disagree (internal error)")
            pos_db[sel] = rp[a:b][idx]
        res.free()
        if filter_ == "mif0":
            fv_p, order_p = db.filter_values(), db.filter_order()
        else:
            fv_p = (dbfile.splitmix_unit(keys_db) if db.num_keys else np.zeros(0)).astype(np.float32)
            order_p = np.argsort(dbfile.filter_sort_code(fv_p, keys_db), kind="stable")

        def write_shard(file):
            dbfile.write_db_positions(file, seq_name, tree_index, newick, k, omega, keys_db, db.key_offsets(), br_db, sc_db, pos_db, fv_p, order_p)
    elif filter_ == "mif0":
        # records packed on the device in filter order and streamed to the file (ipkgpu_db_write): the database itself on one
        # GPU, this rank's shard on several (a shard's header carries only its totals)
        def write_shard(file):
            if world == 1:
                dbfile.write_db_device(eng, db, file, seq_name, tree_index, newick, k, omega)
            else:
                dbfile.write_db_device(eng, db, file, seq_name, [], "", k, omega)
    else:
        # random_filter (filter.cpp:122-145) draws uniform(0, 1) from std::default_random_engine(42) in the hash map's
        # iteration order, which no other build reproduces; here: one fixed draw per k-mer CODE, so the file does not
        # depend on how the k-mers are sharded
        fv = (dbfile.splitmix_unit(db.keys()) if db.num_keys else np.zeros(0)).astype(np.float32)
        order = np.argsort(dbfile.filter_sort_code(fv, db.keys()), kind="stable")
        br, sc = db.entries()

        def write_shard(file):
            one = world == 1
            dbfile.write_db(file, seq_name, tree_index if one else [], newick if one else "", k, omega, db.keys(), db.key_offsets(),
                            br, sc, fv, order)
    totals = distributed.write_db_file(output, seq_name, tree_index, newick, k, omega, write_shard, workdir, dist, world, rank)
    if world == 1:
        totals = (db.num_keys, db.num_entries)
    t_write = time.time() - t0
    emitted = parts.emitted
    if world > 1:
        import torch
        e = torch.tensor([emitted], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(e)
        emitted = int(e.item())
    if verbosity and rank == 0:
        # the reference prints the same three stage timers (db_builder.cpp:236,290,336)
        click.echo(f"Loaded {len(labels)} node matrices ({arp.sites} sites) in {t_load * 1e3:.0f} ms" + (f" on each of {world} ranks" if world > 1 else ""))
        click.echo(f"Computation time: {t_score * 1e3:.0f} ms ({emitted} scored phylo-k-mers)")
        click.echo(f"Filtering time: {t_filter * 1e3:.0f} ms")
        click.echo(f"Merge time: {t_write * 1e3:.0f} ms")
        click.echo(f"Output: {output} ({totals[0]} k-mers, {totals[1]} entries)")
        click.echo("Note: the database layout is a reconstruction of i2l's Boost binary archive (i2l and Boost are not part of the "
                   "reference tree): UNPINNED against a real .ipk -- ipk_amd/csrc/ipk_format.hpp is the one file that knows the bytes; "
                   "IPKGPU_BOOST_ARCHIVE_VERSION sets the archive's library version (default 19).  Guessed fields: the protocol-version "
                   "word behind the archive preamble and the positions flag behind the sequence type (position, width, value; "
                   f"IPKGPU_IPK_PROTOCOL_VERSION, now {dbfile.protocol_version()}, 0 = both left out), the widths of the tree index "
                   "(u64 / f64), filter value (f32) and key (u32)" + (", and the u16 window position of a positioned entry." if keep_positions else "."))
    db.free(); parts.free(); eng.close(); arp.close()
    if world > 1 and own_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    ipk()
