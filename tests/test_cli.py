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
    out = tmp_path / "DB.ipkgpu"
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


def _cli_rank(rank, world, port, tmp, args):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0",
                      IPK_DIST_BACKEND="gloo")            # both ranks on GPU 0, host-staged transport: a rehearsal of torchrun
    res = CliRunner().invoke(cli.ipk, args + ["-o", os.path.join(tmp, "multi.ipkgpu")])
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
    one = tmp_path / "one.ipkgpu"
    res = CliRunner().invoke(cli.ipk, args + ["-o", str(one)])
    assert res.exit_code == 0, res.output
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_cli_rank, args=(2, port, str(tmp_path), args), nprocs=2, join=True)
    assert open(tmp_path / "multi.ipkgpu", "rb").read() == open(one, "rb").read()
    out = open(tmp_path / "rank0.out").read()
    import re
    scored = lambda text: re.search(r"\((\d+) scored phylo-k-mers\)", text).group(1)
    assert "on each of 2 ranks" in out and scored(out) == scored(res.output)
    assert not list((tmp_path / "shards").glob("*.npz"))
