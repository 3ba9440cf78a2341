"""CPU suite (the loader is host code): RAxML-ng .ancestralProbs reader vs its Python restatement."""
import numpy as np
import pytest

import ipk_amd
from ipk_amd.loader import AncestralProbs
from oracle import ar_oracle


def write_probs(path, sigma, labels, sites, seed, fmt="%.9f", extras=True):
    rng = np.random.default_rng(seed)
    states = "ACGT" if sigma == 4 else "ARNDCQEGHILKMFPSTWYV"
    with open(path, "w") as fh:
        fh.write("Node\tSite\tState\t" + "\t".join("p_" + c for c in states) + "\n")
        for li, lab in enumerate(labels):
            p = rng.dirichlet(np.full(sigma, 0.1), size=sites)
            for s in range(sites):
                vals = [fmt % v for v in p[s]]
                if extras and s == 3:
                    vals[0] = "%.6e" % p[s][0]                  # scientific notation
                    vals[1] = " " + vals[1] + " "               # padded field (trim_chars<' '>)
                if extras and s == 5:
                    vals[2] = "0"                               # zero probability -> log10 = -inf
                fh.write(f"{lab}\t{s + 1}\t{states[int(np.argmax(p[s]))]}\t" + "\t".join(vals) + "\n")
            if extras and li == 0:
                fh.write("\n.comment line between nodes\n")


@pytest.mark.parametrize("sigma", [4, 20])
def test_loader_matches_restatement(tmp_path, sigma):
    path = tmp_path / "x.raxml.ancestralProbs"
    labels = ["Node1", "Node2", "Node10", "n_X0", "n_X1"]
    write_probs(path, sigma, labels, 12, 5 + sigma)
    ar = AncestralProbs(path, sigma)
    assert ar.labels == labels and ar.sites == 12
    mats = ar.read()
    ref, order = ar_oracle.read_file(path, sigma)
    assert order == labels
    for i, lab in enumerate(labels):
        assert np.array_equal(mats[i].view(np.uint32), ref[lab].view(np.uint32)), lab
    # a subset, in caller order, multi-threaded
    sub = ar.read(["n_X1", "Node2"], n_threads=3)
    assert np.array_equal(sub[0], mats[4]) and np.array_equal(sub[1], mats[1])
    assert np.isneginf(mats[:, 5, 2 if sigma == 4 else ar_oracle.AA_IPK_FROM_RAXML.index(2)]).all()
    # csv-parser's float accumulation is close to, but not always equal to, a correctly rounded strtof
    strtof = np.array([[float(v) for v in line.split("\t")[3:]] for line in open(path).read().splitlines()[1:13]],
                      dtype=np.float32)
    if sigma == 20:
        strtof = strtof[:, ar_oracle.AA_IPK_FROM_RAXML]
    with np.errstate(divide="ignore"):
        assert np.allclose(mats[0][np.isfinite(mats[0])], np.log10(strtof)[np.isfinite(mats[0])], rtol=0, atol=2e-6)
    ar.close()


def test_loader_errors(tmp_path):
    with pytest.raises(ipk_amd.IpkGpuError):
        AncestralProbs(tmp_path / "missing", 4)
    path = tmp_path / "bad.raxml.ancestralProbs"
    write_probs(path, 4, ["A", "B"], 6, 1, extras=False)
    txt = open(path).read().splitlines()
    txt[9] = txt[9].rsplit("\t", 1)[0]                       # node B loses a column in one row
    open(path, "w").write("\n".join(txt) + "\n")
    ar = AncestralProbs(path, 4)
    with pytest.raises(KeyError):
        ar.read(["nope"])
    with pytest.raises(ipk_amd.IpkGpuError) as ei:
        ar.read(["B"])
    assert "Could not read the AR matrix for the node B" in str(ei.value)
    ar.close()


def test_loader_feeds_the_oracle_pipeline(tmp_path):
    """Loader output has the layout the scoring path takes (site-major log10 float32)."""
    from oracle import ipk_oracle as co
    path = tmp_path / "y.raxml.ancestralProbs"
    write_probs(path, 4, ["g_X0", "g_X1"], 30, 9, extras=False)
    mats = AncestralProbs(path, 4).read()
    keys, scores, emitted = co.explore_group(mats, 6, co.log_threshold(1.5, 4, 6))
    assert emitted > 0 and len(keys) > 0 and np.all(scores <= 0)


def test_index_is_stitched_across_thread_ranges(tmp_path):
    """A file large enough for the index scan to be split over several threads (8 MB of text per range): labels, their
    order and every block boundary must come out as from one left-to-right pass -- including a node whose block spans a
    cut and a label that re-appears later in the file (the later block wins, ar.cpp:181 assigns into the map)."""
    rng = np.random.default_rng(3)
    rows = ["\t".join("%.9f" % v for v in rng.dirichlet(np.full(4, 0.1))) for _ in range(997)]
    n_nodes, sites = 140, 2500
    labels = [f"Node{i + 1}" for i in range(n_nodes)]
    path = tmp_path / "big.raxml.ancestralProbs"
    with open(path, "w") as fh:
        fh.write("Node\tSite\tState\tp_A\tp_C\tp_G\tp_T\n")
        for li, lab in enumerate(labels):
            fh.write("".join(f"{lab}\t{s + 1}\tA\t{rows[(li * 31 + s * 7) % 997]}\n" for s in range(sites)))
        fh.write("".join(f"Node3\t{s + 1}\tC\t{rows[(5 + s) % 997]}\n" for s in range(sites)))      # Node3 again: this block counts
    assert path.stat().st_size > 20 << 20
    ar = AncestralProbs(path, 4)
    assert ar.labels == labels and ar.sites == sites
    pick = ["Node1", "Node3", "Node70", "Node71", "Node140"]
    mats = ar.read(pick, n_threads=4)
    for i, lab in enumerate(pick):
        li = labels.index(lab)
        src = [rows[(5 + s) % 997] if lab == "Node3" else rows[(li * 31 + s * 7) % 997] for s in range(sites)]
        ref = np.array([[ar_oracle.parse_float(v) for v in r.split("\t")] for r in src], dtype=np.float32)
        from oracle import ipk_oracle as co
        assert np.array_equal(mats[i].view(np.uint32), co.log10f(ref).view(np.uint32)), lab
    ar.close()
