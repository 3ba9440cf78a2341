"""CPU suite (host code): reference tree, ghost nodes, AR node mapping (row n3) against the Python restatement."""
import numpy as np
import pytest

from ipk_amd import tree as T
from oracle import tree_oracle as to


def random_newick(rng, n_leaves, rooted=True):
    nodes = [f"t{i}:{rng.uniform(0.01, 0.9):.6f}" for i in range(n_leaves)]
    inner = 0
    while len(nodes) > (2 if rooted else 3):
        i, j = sorted(rng.choice(len(nodes), size=2, replace=False))
        b, a = nodes.pop(j), nodes.pop(i)
        label = f"n{inner}" if rng.random() < 0.5 else ""
        inner += 1
        nodes.append(f"({a},{b}){label}:{rng.uniform(0.01, 0.9):.6f}")
    return "(" + ",".join(nodes) + ")root;"


@pytest.mark.parametrize("seed,n_leaves,rooted", [(1, 2, True), (2, 5, True), (3, 9, True), (4, 30, True), (5, 12, False), (6, 3, False)])
def test_tree_extension_and_plan_match_the_restatement(seed, n_leaves, rooted):
    rng = np.random.default_rng(seed)
    nw = random_newick(rng, n_leaves, rooted)
    ot, root = T.Tree.parse(nw), to.parse(nw)
    post = to.postorder(root)
    assert ot.num_nodes == len(post) and ot.is_rooted == (len(root.children) == 2) and ot.num_leaves == root.num_leaves
    assert ot.labels() == [n.label for n in post]
    nn, sl = ot.index()
    assert nn.tolist() == [n.num_nodes for n in post]
    assert np.allclose(sl, [n.subtree_length for n in post], rtol=1e-12, atol=0)
    # a re-parse of the serialised tree is the same tree
    again = T.Tree.parse(ot.newick())
    assert again.labels() == ot.labels() and [again.branch_length(i) for i in range(again.num_nodes)] == [ot.branch_length(i) for i in range(ot.num_nodes)]
    # extension: names, topology (labels in post-order + parents), branch lengths
    et = ot.extend()
    eroot, mapping = to.extend(root)
    epost = to.postorder(eroot)
    assert et.num_nodes == len(epost) == ot.num_nodes + 4 * (ot.num_nodes - 1)
    assert et.labels() == [n.label for n in epost]
    assert [et.parent(i) for i in range(et.num_nodes)] == [(-1 if n.parent is None else n.parent.postorder) for n in epost]
    assert np.allclose([et.branch_length(i) for i in range(et.num_nodes)], [n.length for n in epost], rtol=1e-12, atol=0)
    # AR tree as RAxML-ng returns it (unrooted, own labels), rerooted, mapped in lock step
    relabel = lambda n: n.label if not n.children else f"Node{n.postorder + 1}"
    ar_nw = to.to_unrooted_ar(eroot, relabel)
    art, ar_root = T.Tree.parse(ar_nw), to.parse(ar_nw)
    if rooted:
        assert not art.is_rooted
        art.reroot()
        ar_root = to.reroot(ar_root)
        assert art.label(art.num_nodes - 1) == "added_root"
    amap = to.map_nodes(eroot, ar_root)
    for strategy in ("both", "inner-only", "outer-only"):
        want = [(lab, amap[lab], b) for b, labs in to.ghost_groups(root, eroot, mapping, strategy) for lab in labs]
        assert T.ghost_plan(ot, et, art, strategy) == want
        # every ghost maps to the AR label of the same node (relabelled inner node)
        assert all(ar == f"Node{next(n for n in epost if n.label == lab).postorder + 1}" for lab, ar, _ in want)
    # every non-root branch has its group (both ghosts), root has none
    plan = T.ghost_plan(ot, et, None, "both")
    assert sorted({b for _, _, b in plan}) == [i for i in range(ot.num_nodes - 1)]
    assert all(ext == ar for ext, ar, _ in plan)


def test_newick_details_and_errors():
    t = T.Tree.parse(" ( 'a b':0.5 , [comment] B:1e-1 , (C,D)x:2 ) ; ")
    assert t.labels() == ["a b", "B", "C", "D", "x", ""] and not t.is_rooted and t.branch_length(1) == 0.1
    with pytest.raises(Exception):
        T.Tree.parse("((A,B);")
    with pytest.raises(Exception):
        T.Tree.parse("(A,B)")
    with pytest.raises(Exception):
        T.Tree.load("/nonexistent/tree.nwk")
    o = T.Tree.parse("((A:1,B:1):1,C:2);")
    e = o.extend()
    bad = T.Tree.parse("(A,B,C);")
    with pytest.raises(Exception):
        T.ghost_plan(o, e, bad, "both")                   # node counts differ (ar.cpp:792-798)
    assert T.ghost_plan(o, o, None, "both") == []         # no ghost labels: an empty plan
