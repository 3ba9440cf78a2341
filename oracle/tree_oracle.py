"""Python restatement of IPK's tree handling -- TEST INFRASTRUCTURE ONLY (checks ipk_amd/csrc/phylo_host.cpp).

Follows, function by function (paths relative to the IPK tree):
  ipk/src/extended_tree.cpp:7-33     total_branch_length
  ipk/src/extended_tree.cpp:36-73    calc_ghost_branch_lengths
  ipk/src/extended_tree.cpp:76-150   tree_extender (recursive extend_subtree, counter from node_count + 1)
  ipk/src/extended_tree.cpp:186-205  reroot_tree
  ipk/src/ar.cpp:790-834             map_nodes
  ipk/src/db_builder.cpp:495-553     is_ghost / get_ghost_ids / group_ghost_ids
  ipk/src/db_builder.cpp:192-197     tree index
i2l::phylo_tree is un-vendored; assumed: children in newick order, post-order ids from 0, iteration = post-order,
get_subtree_branch_length = branches below the node.
"""


class N:
    def __init__(self, label="", length=0.0, parent=None):
        self.label, self.length, self.parent, self.children = label, length, parent, []
        self.postorder = self.num_nodes = self.num_leaves = 0
        self.subtree_length = 0.0

    def add_child(self, c):
        c.parent = self
        self.children.append(c)

    def remove_child(self, c):
        self.children.remove(c)


def parse(newick):
    s = newick.strip()
    pos = 0

    def node(parent):
        nonlocal pos
        n = N(parent=parent)
        if s[pos] == "(":
            pos += 1
            while True:
                n.children.append(node(n))
                if s[pos] == ",":
                    pos += 1
                    continue
                assert s[pos] == ")"
                pos += 1
                break
        j = pos
        while s[j] not in "(),:;":
            j += 1
        n.label = s[pos:j]
        pos = j
        if s[pos] == ":":
            j = pos + 1
            while s[j] not in "(),;":
                j += 1
            n.length = float(s[pos + 1:j])
            pos = j
        return n

    root = node(None)
    assert s[pos] == ";"
    index(root)
    return root


def postorder(root):
    out = []

    def rec(n):
        for c in n.children:
            rec(c)
        out.append(n)
    rec(root)
    return out


def index(root):
    for i, n in enumerate(postorder(root)):
        n.postorder = i
        n.num_nodes = 1 + sum(c.num_nodes for c in n.children)
        n.num_leaves = 1 if not n.children else sum(c.num_leaves for c in n.children)
        n.subtree_length = sum(c.subtree_length + c.length for c in n.children)


def copy(root):
    def rec(n, parent):
        m = N(n.label, n.length, parent)
        m.postorder, m.num_nodes, m.num_leaves, m.subtree_length = n.postorder, n.num_nodes, n.num_leaves, n.subtree_length
        m.children = [rec(c, m) for c in n.children]
        return m
    return rec(root, None)


def total_branch_length(root):
    if not root.children:
        return 0.0
    length = 0.0
    for n in postorder(root):
        length += n.length if not n.children else n.num_leaves * n.length
    return length - root.num_leaves * root.length


def ghost_lengths(node):
    old = node.length
    x0 = old / 2.0
    residual = old - x0
    if not node.children:
        return x0, residual
    return x0, (total_branch_length(node) + residual * node.num_leaves) / node.num_leaves


def extend(original_root):
    by_post = {n.postorder: n for n in postorder(original_root)}
    ext = copy(original_root)
    counter = [len(by_post) + 1]
    mapping = {}

    def extend_subtree(node):
        for child in list(node.children):
            extend_subtree(child)
        if node.parent is not None:
            parent = node.parent
            x0_len, x1_len = ghost_lengths(by_post[node.postorder])
            x0_name = f"{counter[0]}_X0"; counter[0] += 1
            x0 = N(x0_name, x0_len)
            parent.remove_child(node)
            parent.add_child(x0)
            x1_name = f"{counter[0]}_X1"; counter[0] += 1
            x1 = N(x1_name, x1_len)
            x0.add_child(x1)
            x0.add_child(node)
            node.length = node.length - x0_len
            for suffix in ("_X2", "_X3"):
                x1.add_child(N(f"{counter[0]}{suffix}", 0.01)); counter[0] += 1
            mapping[x0_name] = node.postorder
            mapping[x1_name] = node.postorder

    extend_subtree(ext)
    index(ext)
    return ext, mapping


def reroot(root):
    if len(root.children) > 2:
        a = root.children[0]
        new = N("added_root", 0.0)
        new.add_child(root)
        new.add_child(a)
        root.remove_child(a)
        index(new)
        return new
    return root


def map_nodes(ext_root, ar_root):
    e, a = postorder(ext_root), postorder(ar_root)
    assert len(e) == len(a)
    return {x.label: y.label for x, y in zip(e, a) if x.label}


def ghost_groups(original_root, ext_root, mapping, strategy="both"):
    """[(branch id, [ghost labels])] in first-seen order (db_builder.cpp:495-553)."""
    suffixes = {"both": ("_X0", "_X1"), "inner-only": ("_X0",), "outer-only": ("_X1",)}[strategy]
    groups, where = [], {}
    for n in postorder(ext_root):
        if not n.label.endswith(suffixes):
            continue
        b = mapping[n.label]
        if b == original_root.postorder:
            continue
        if b in where:
            groups[where[b]][1].append(n.label)
        else:
            where[b] = len(groups)
            groups.append((b, [n.label]))
    return groups


def to_unrooted_ar(ext_root, relabel):
    """What RAxML-ng hands back for a rooted input ((A,B)n,C)r: the unrooted (C,A,B) with its own inner labels
    (main.cpp:172-178).  `relabel(node)` gives the AR label.  Returns a newick string."""
    def nw(n):
        inner = "(" + ",".join(nw(c) for c in n.children) + ")" if n.children else ""
        return f"{inner}{relabel(n)}:{n.length!r}"
    if len(ext_root.children) == 2:
        left, right = ext_root.children
        assert len(left.children) == 2
        parts = [nw(right)] + [nw(c) for c in left.children]
        return "(" + ",".join(parts) + ")" + relabel(left) + ";"
    return "(" + ",".join(nw(c) for c in ext_root.children) + ")" + relabel(ext_root) + ";"
