"""CLI (row n4): option surface on CPU, end-to-end build on the GPU."""
import numpy as np
import pytest
from click.testing import CliRunner

from ipk_amd import cli, dbfile
from oracle import db_oracle as dbo
from oracle import ipk_oracle as co
from tests.test_loader import write_probs


def test_cli_options_match_reference_surface():
    # option names of `ipk.py build` (ipk.py:70-202) that a user script may pass
    names = {o for p in cli.build.params for o in p.opts}
    for opt in ["-b", "--ar", "-r", "--refalign", "-t", "--reftree", "-s", "--states", "-v", "--verbosity", "-w", "--workdir",
                "--write-reduction", "-a", "--alpha", "-c", "--categories", "-k", "--k", "-m", "--model", "--convert-uo",
                "--no-reduction", "--reduction-ratio", "--omega", "--filter", "-u", "--mu", "--ghosts", "--use-unrooted",
                "--merge-branches", "--ar-dir", "--ar-only", "--ar-config", "--keep-positions", "--uncompressed", "--threads",
                "--output", "-o", "--on-disk"]:
        assert opt in names, opt
    defaults = {p.name: p.default for p in cli.build.params}
    assert defaults["k"] == 8 and defaults["omega"] == 1.5 and defaults["reduction_ratio"] == 0.99
    assert defaults["filter_"] == "mif0" and defaults["ghosts"] == "both" and defaults["states"] == "nucl"
    res = CliRunner().invoke(cli.ipk, ["build", "--help"])
    assert res.exit_code == 0 and "--ar-dir" in res.output


@pytest.mark.gpu
def test_cli_build_end_to_end(tmp_path):
    ar_dir = tmp_path / "AR"; ar_dir.mkdir()
    labels = [f"{i}_X{j}" for i in range(4) for j in range(2)]
    write_probs(ar_dir / "ar.raxml.ancestralProbs", 4, labels, 40, 11, extras=False)
    with open(tmp_path / "map.tsv", "w") as fh:
        for i, lab in enumerate(labels):
            fh.write(f"{lab}\t{7 + i // 2}\n")
    out = tmp_path / "DB.ipk"
    res = CliRunner().invoke(cli.ipk, ["build", "-w", str(tmp_path), "--ar-dir", str(ar_dir), "--mapping", str(tmp_path / "map.tsv"),
                                       "-k", "6", "--omega", "1.5", "-o", str(out), "--num-tree-nodes", "9"])
    assert res.exit_code == 0, res.output
    assert "Computation time" in res.output and "Filtering time" in res.output
    hdr, recs = dbfile.read_db(out)
    # against the oracle pipeline on the same file
    from oracle import ar_oracle
    mats, order = ar_oracle.read_file(ar_dir / "ar.raxml.ancestralProbs", 4)
    eps = co.log_threshold(1.5, 4, 6)
    full = dbo.build_db([(7 + g,) + co.explore_group(np.stack([mats[labels[2 * g]], mats[labels[2 * g + 1]]]), 6, eps)[:2]
                         for g in range(4)])
    assert hdr["total_num_kmers"] == len(full) and hdr["kmer_size"] == 6
    thr = co.score_threshold(1.5, 4, 6)
    for key, fv, br, sc in recs[:200]:
        assert [(int(b), int(s)) for b, s in zip(br, sc.view(np.uint32))] == full[key]
        ref = co.mif0(np.array([s for _, s in full[key]], dtype=np.uint32).view(np.float32), 9, thr)
        assert abs(fv - ref) <= 1e-6 * max(1.0, abs(ref))
    assert all(recs[i][1] <= recs[i + 1][1] for i in range(len(recs) - 1))


@pytest.mark.gpu
def test_cli_keep_positions(tmp_path):
    """`ipk.py build --keep-positions --states amino` (ipk-aa-pos): every entry of the database carries the window position that goes
    with its kept score -- the larger score wins, equal scores keep the EARLIER window (branch_group.cpp:73-86) -- checked against the
    oracle's positioned groups; DNA is refused as the reference's wrapper refuses it (ipk.py:281-282)."""
    ar_dir = tmp_path / "AR"; ar_dir.mkdir()
    labels = [f"{i}_X{j}" for i in range(3) for j in range(2)]
    write_probs(ar_dir / "ar.raxml.ancestralProbs", 20, labels, 30, 5, extras=False)
    with open(tmp_path / "map.tsv", "w") as fh:
        for i, lab in enumerate(labels):
            fh.write(f"{lab}\t{4 + i // 2}\n")
    out = tmp_path / "DBpos.ipk"
    res = CliRunner().invoke(cli.ipk, ["build", "-w", str(tmp_path), "--ar-dir", str(ar_dir), "--mapping", str(tmp_path / "map.tsv"), "-s", "amino",
                                       "-k", "4", "--omega", "1.5", "-o", str(out), "--num-tree-nodes", "7", "--keep-positions"])
    assert res.exit_code == 0, (res.output, res.exception)
    hdr, recs = dbfile.read_db(out)
    assert hdr["positions_loaded"] is True and hdr["sequence_type"] == "AA"
    from oracle import ar_oracle
    mats, _ = ar_oracle.read_file(ar_dir / "ar.raxml.ancestralProbs", 20)
    eps = co.log_threshold(1.5, 20, 4)
    full = {}
    for g in range(3):
        keys, scores, pos, _ = co.explore_group_pos(np.stack([mats[labels[2 * g]], mats[labels[2 * g + 1]]]), 4, eps)
        for kk, sc, pp in zip(keys.tolist(), scores.view(np.uint32).tolist(), pos.tolist()):
            full.setdefault(kk, []).append((4 + g, sc, pp))
    assert hdr["total_num_kmers"] == len(full) and hdr["total_num_entries"] == sum(len(v) for v in full.values())
    for key, fv, br, sc, pos in recs:
        assert [(int(b), int(s), int(p)) for b, s, p in zip(br, sc.view(np.uint32), pos)] == full[key]
    assert all(recs[i][1] <= recs[i + 1][1] for i in range(len(recs) - 1))
    bad = CliRunner().invoke(cli.ipk, ["build", "-w", str(tmp_path), "--ar-dir", str(ar_dir), "--mapping", str(tmp_path / "map.tsv"),
                                       "-k", "4", "-o", str(out), "--keep-positions"])
    assert bad.exit_code != 0 and "not supported for DNA" in bad.output


def _cli_rank(rank, world, port, tmp, args):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                      IPK_DIST_BACKEND="gloo")            # both ranks on GPU 0, host-staged transport: a rehearsal of torchrun
    res = CliRunner().invoke(cli.ipk, args + ["-o", os.path.join(tmp, "multi.ipk")])
    assert res.exit_code == 0, (res.output, res.exception)
    if rank == 0:
        open(os.path.join(tmp, "rank0.out"), "w").write(res.output)


@pytest.mark.gpu
@pytest.mark.parametrize("filt", ["mif0", "random"])
def test_cli_build_two_ranks_writes_the_same_file(tmp_path, filt):
    """`torchrun --nproc-per-node 2 -m ipk_amd.cli build ...`: groups sharded over the ranks, k-mer-keyed exchange, filter
    values per shard, shard files merged by filter value -- byte-identical to the single-GPU file."""
    import socket
    import torch.multiprocessing as mp
    ar_dir = tmp_path / "AR"; ar_dir.mkdir()
    labels = [f"{i}_X{j}" for i in range(5) for j in range(2)]
    write_probs(ar_dir / "ar.raxml.ancestralProbs", 4, labels, 60, 21, extras=False)
    with open(tmp_path / "map.tsv", "w") as fh:
        for i, lab in enumerate(labels):
            fh.write(f"{lab}\t{3 + i // 2}\n")
    args = ["build", "-w", str(tmp_path), "--ar-dir", str(ar_dir), "--mapping", str(tmp_path / "map.tsv"), "-k", "7",
            "--filter", filt, "--num-tree-nodes", "11"]
    one = tmp_path / "one.ipk"
    res = CliRunner().invoke(cli.ipk, args + ["-o", str(one)])
    assert res.exit_code == 0, res.output
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_cli_rank, args=(2, port, str(tmp_path), args), nprocs=2, join=True)
    assert open(tmp_path / "multi.ipk", "rb").read() == open(one, "rb").read()
    out = open(tmp_path / "rank0.out").read()
    import re
    scored = lambda text: re.search(r"\((\d+) scored phylo-k-mers\)", text).group(1)
    assert "on each of 2 ranks" in out and scored(out) == scored(res.output)
    assert not list((tmp_path / "shards").glob("*.npz"))


def _reference_workdir(tmp_path, n_leaves, sites, seed, rooted=True):
    """A workdir holding ONLY reference-format artefacts: the reference tree (newick) and, in AR/, what RAxML-ng leaves
    behind for the extended tree (`*.raxml.ancestralProbs` with one block per inner node, `*.raxml.ancestralTree`:
    unrooted, RAxML's own inner labels).  Returns (tree path, AR dir, restatement objects for the checks)."""
    from oracle import tree_oracle as to
    from tests.test_tree import random_newick
    rng = np.random.default_rng(seed)
    nw = random_newick(rng, n_leaves, rooted)
    tree_file = tmp_path / "tree.newick"
    tree_file.write_text(nw + "\n")
    root = to.parse(nw)
    eroot, mapping = to.extend(root)
    relabel = lambda n: n.label if not n.children else f"Node{n.postorder + 1}"
    ar_dir = tmp_path / "AR"; ar_dir.mkdir()
    (ar_dir / "extended_align.phylip.raxml.ancestralTree").write_text(to.to_unrooted_ar(eroot, relabel) + "\n")
    inner = [relabel(n) for n in to.postorder(eroot) if n.children]
    write_probs(ar_dir / "extended_align.phylip.raxml.ancestralProbs", 4, inner, sites, seed + 1, extras=False)
    return tree_file, ar_dir, (root, eroot, mapping, relabel)


@pytest.mark.gpu
@pytest.mark.parametrize("n_leaves,sites,k,ghosts", [(8, 64, 8, "both"),          # BASELINE configs[0]: tiny tree, 64 sites, k=8
                                                     (11, 1400, 10, "both"),      # configs[4] stand-in: D652-like width, 20 branches -> 40 ghost nodes
                                                     (6, 50, 7, "inner-only")])
def test_cli_builds_from_a_reference_workdir(tmp_path, n_leaves, sites, k, ghosts):
    """`ipk.py build -r -t -w -k --ar-dir` with nothing but reference-format inputs: tree extension, ghost naming, AR node
    mapping and grouping all derived here (no --mapping), file checked against the oracle pipeline on the same inputs."""
    from oracle import ar_oracle, tree_oracle as to
    tree_file, ar_dir, (root, eroot, mapping, relabel) = _reference_workdir(tmp_path, n_leaves, sites, 100 + n_leaves)
    work = tmp_path / "work"
    out = tmp_path / "DB.ipk"
    (tmp_path / "aln.fasta").write_text(">t0\nACGT\n")                       # -r is accepted and not read here
    res = CliRunner().invoke(cli.ipk, ["build", "-r", str(tmp_path / "aln.fasta"), "-t", str(tree_file), "-w", str(work), "-k", str(k),
                                       "--ar-dir", str(ar_dir), "--ghosts", ghosts, "-o", str(out)])
    assert res.exit_code == 0, (res.output, res.exception)
    assert (work / "extended_trees" / "extended_tree.newick").exists()
    hdr, (keys, fvs, counts, eoff, br, sc) = dbfile.read_db(out, as_arrays=True)
    post = to.postorder(root)
    assert hdr["kmer_size"] == k and hdr["sequence_type"] == "DNA" and len(hdr["tree_index"]) == len(post)
    assert [t[0] for t in hdr["tree_index"]] == [n.num_nodes for n in post]
    assert to.postorder(to.parse(hdr["newick"]))[-1].num_nodes == len(post)
    # the oracle pipeline: plan from the restatement, matrices from the Python reader, explore_group per branch
    ar_root = to.reroot(to.parse(to.to_unrooted_ar(eroot, relabel)))
    amap = to.map_nodes(eroot, ar_root)
    mats, _ = ar_oracle.read_file(ar_dir / "extended_align.phylip.raxml.ancestralProbs", 4)
    eps = co.log_threshold(1.5, 4, k)
    ks, bs, ss = [], [], []
    for gi, (branch, labs) in enumerate(to.ghost_groups(root, eroot, mapping, ghosts)):
        gk, gs, _ = co.explore_group(np.stack([mats[amap[lab]] for lab in labs]), k, eps)
        ks.append(gk.astype(np.int64)); ss.append(gs.view(np.uint32)); bs.append(np.full(len(gk), branch, np.uint32))
    key, bra, bits = np.concatenate(ks), np.concatenate(bs), np.concatenate(ss)
    gidx = np.concatenate([np.full(len(x), i) for i, x in enumerate(ks)])
    o = np.lexsort((gidx, key))                                            # per key: entries in group order (db_builder.cpp:685-694)
    key, bra, bits = key[o], bra[o], bits[o]
    uk, first = np.unique(key, return_index=True)
    assert hdr["total_num_kmers"] == len(uk) and hdr["total_num_entries"] == len(key)
    # file order: ascending filter value, ties by key; compare per key through the key's position in the oracle arrays
    assert np.all(np.diff(fvs) >= 0)
    pos = np.searchsorted(uk, keys)
    assert np.array_equal(uk[pos], keys) and len(np.unique(keys)) == len(keys)
    ends = np.concatenate([first[1:], [len(key)]])
    assert np.array_equal(counts, (ends - first)[pos].astype(np.uint64))
    # entries of all k-mers, in file order
    src = np.concatenate([np.arange(first[p], ends[p]) for p in pos[:20000]])
    n_chk = int(eoff[min(20000, len(keys))])
    assert np.array_equal(br[:n_chk], bra[src]) and np.array_equal(sc[:n_chk].view(np.uint32), bits[src])
    thr = co.score_threshold(1.5, 4, k)
    for i in np.linspace(0, len(keys) - 1, 300).astype(np.int64):
        p = pos[i]
        ref = co.mif0(bits[first[p]:ends[p]].view(np.float32), len(post), thr)
        assert abs(fvs[i] - ref) <= 1e-6 * max(1.0, abs(ref))


def test_ipk_py_entry_point_exists_and_forwards():
    """The reference's command line starts with `ipk.py build` (ipk.py:203-230); the repository root carries that file."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "ipk.py"), "build", "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--reftree" in r.stdout and "--ar-dir" in r.stdout and "--omega" in r.stdout
