// phylo_host.cpp -- host side of the group loop's inputs: reference tree, ghost nodes, node mapping (C ABI: ipkgpu_tree_*,
// ipkgpu_ghost_plan_*; declared in include/ipkgpu.h).
//
// What db_builder::explore_kmers needs before it can score anything (citations relative to the IPK tree):
//   ipk/src/extended_tree.cpp:76-162   tree_extender: ghost nodes X0/X1 (+ dummy leaves X2/X3) on every non-root branch,
//                                      named <counter>_X0 .. with counter starting at node_count + 1; ghost -> original
//                                      post-order id mapping
//   ipk/src/extended_tree.cpp:186-205  reroot_tree: (a, b, c); -> ((b, c), a)added_root;
//   ipk/src/ar.cpp:790-834             map_nodes: extended tree and AR tree walked in lock step, post-order
//   ipk/src/db_builder.cpp:495-553     is_ghost / get_ghost_ids / group_ghost_ids: ghost labels by strategy in tree
//                                      iteration order, grouped by original post-order id in first-seen order, root skipped
//   ipk/src/db_builder.cpp:192-197     tree index: (num_nodes, subtree_branch_length) per node of the original tree
//
// i2l::phylo_tree itself (newick reader, node indexing) is un-vendored; this is an own small tree with the semantics the
// call sites above rely on: children kept in newick order, post-order ids from 0 (children before their parent), tree
// iteration = post-order.  Assumptions about i2l that no reference file pins are marked ASSUMPTION.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ipkgpu.h"

namespace {

struct Node {
    std::string label;
    double length = 0.0;
    bool has_length = false;
    int parent = -1;
    std::vector<int> children;
    // index(): post-order id, size of the subtree, leaves below, total branch length below the node
    uint32_t postorder = 0, num_nodes = 1, num_leaves = 0;
    double subtree_length = 0.0;
};

thread_local std::string g_tree_err;

}  // namespace

struct ipkgpu_tree {
    std::vector<Node> nodes;          // arena; ids are arena indices
    int root = -1;
    std::vector<int> post;            // arena index of every node in post-order
    std::unordered_map<std::string, uint32_t> ghost_to_branch;   // filled by extend(): ghost label -> original post-order id
    std::string newick_cache;

    int add(const std::string& label, double len, bool has_len, int parent)
    {
        Node n; n.label = label; n.length = len; n.has_length = has_len; n.parent = parent;
        nodes.push_back(std::move(n));
        const int id = (int)nodes.size() - 1;
        if (parent >= 0) nodes[parent].children.push_back(id);
        return id;
    }
    void remove_child(int parent, int child)
    {
        auto& c = nodes[parent].children;
        for (size_t i = 0; i < c.size(); ++i) if (c[i] == child) { c.erase(c.begin() + i); return; }
    }
    // phylo_tree::index(): iterative post-order
    void index()
    {
        post.clear();
        if (root < 0) return;
        std::vector<std::pair<int, size_t>> st;
        st.push_back({root, 0});
        while (!st.empty()) {
            auto& [id, next] = st.back();
            if (next < nodes[id].children.size()) { const int c = nodes[id].children[next++]; st.push_back({c, 0}); continue; }
            Node& n = nodes[id];
            n.postorder = (uint32_t)post.size();
            n.num_nodes = 1; n.num_leaves = n.children.empty() ? 1 : 0; n.subtree_length = 0.0;
            for (int c : n.children) {
                n.num_nodes += nodes[c].num_nodes; n.num_leaves += nodes[c].num_leaves;
                // ASSUMPTION (i2l phylo_node::get_subtree_branch_length): the branches BELOW the node, the node's own
                // branch excluded -- the convention of total_branch_length()'s correction at extended_tree.cpp:27-30
                n.subtree_length += nodes[c].subtree_length + nodes[c].length;
            }
            post.push_back(id);
            st.pop_back();
        }
    }
    bool is_rooted() const { return root >= 0 && nodes[root].children.size() == 2; }   // ASSUMPTION: i2l's definition
};

struct ipkgpu_ghost_plan {
    std::vector<std::string> ext_labels, ar_labels;
    std::vector<uint32_t> branches;
    std::vector<uint32_t> tree_num_nodes;
    std::vector<double> tree_subtree_length;
};

namespace {

// ---- newick ---------------------------------------------------------------------------------------
struct Parser {
    const char* s; size_t n, i = 0; ipkgpu_tree* t;
    void skip()
    {
        for (;;) {
            while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
            if (i < n && s[i] == '[') { while (i < n && s[i] != ']') ++i; if (i < n) ++i; continue; }   // comment
            break;
        }
    }
    bool fail(const char* what) { g_tree_err = std::string("newick: ") + what + " at offset " + std::to_string(i); return false; }
    bool label_and_length(int id)
    {
        skip();
        std::string lab;
        if (i < n && (s[i] == '\'' || s[i] == '"')) {
            const char q = s[i++];
            while (i < n && s[i] != q) lab.push_back(s[i++]);
            if (i >= n) return fail("unterminated quoted label");
            ++i;
        } else {
            while (i < n && !strchr("(),:;[ \t\r\n", s[i])) lab.push_back(s[i++]);
        }
        t->nodes[id].label = lab;
        skip();
        if (i < n && s[i] == ':') {
            ++i; skip();
            char* end = nullptr;
            const double v = strtod(s + i, &end);
            if (end == s + i) return fail("branch length expected");
            i = (size_t)(end - s);
            t->nodes[id].length = v; t->nodes[id].has_length = true;
        }
        return true;
    }
    bool subtree(int parent, int& out)
    {
        skip();
        const int id = t->add("", 0.0, false, parent);
        out = id;
        if (i < n && s[i] == '(') {
            ++i;
            for (;;) {
                int c;
                if (!subtree(id, c)) return false;
                skip();
                if (i < n && s[i] == ',') { ++i; continue; }
                if (i < n && s[i] == ')') { ++i; break; }
                return fail("',' or ')' expected");
            }
        }
        return label_and_length(id);
    }
};

void write_newick(const ipkgpu_tree& t, int id, std::string& out)
{
    const Node& n = t.nodes[id];
    if (!n.children.empty()) {
        out.push_back('(');
        for (size_t c = 0; c < n.children.size(); ++c) { if (c) out.push_back(','); write_newick(t, n.children[c], out); }
        out.push_back(')');
    }
    out += n.label;
    if (n.parent >= 0 || n.has_length) {
        // ASSUMPTION: i2l::io::to_newick's number format is unknown; shortest round-trip form of the double
        char buf[40];
        snprintf(buf, sizeof buf, ":%.17g", n.length);
        for (int prec = 1; prec < 17; ++prec) {
            char b2[40];
            snprintf(b2, sizeof b2, ":%.*g", prec, n.length);
            if (strtod(b2 + 1, nullptr) == n.length) { memcpy(buf, b2, sizeof b2); break; }
        }
        out += buf;
    }
}

ipkgpu_tree* clone(const ipkgpu_tree& src)
{
    ipkgpu_tree* t = new ipkgpu_tree();
    t->nodes = src.nodes; t->root = src.root; t->post = src.post;
    return t;
}

// total_branch_length (extended_tree.cpp:7-33): sum over the subtree of (leaves below) x (branch length), the root's own excluded
double total_branch_length(const ipkgpu_tree& t, int root)
{
    if (t.nodes[root].children.empty()) return 0.0;
    double length = 0.0;
    std::vector<int> st{root};
    while (!st.empty()) {
        const int id = st.back(); st.pop_back();
        const Node& n = t.nodes[id];
        length += n.children.empty() ? n.length : n.num_leaves * n.length;
        for (int c : n.children) st.push_back(c);
    }
    return length - t.nodes[root].num_leaves * t.nodes[root].length;
}

bool ends_with(const std::string& s, const char* suf)
{
    const size_t m = strlen(suf);
    return s.size() >= m && memcmp(s.data() + s.size() - m, suf, m) == 0;
}

}  // namespace

extern "C" {

const char* ipkgpu_tree_last_error(void) { return g_tree_err.c_str(); }

int ipkgpu_tree_parse(const char* newick, ipkgpu_tree** out)
{
    if (!newick || !out) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ipkgpu_tree> t(new ipkgpu_tree());
    Parser p{newick, strlen(newick), 0, t.get()};
    int root;
    if (!p.subtree(-1, root)) return IPKGPU_ERR_INVALID;
    p.skip();
    if (p.i >= p.n || newick[p.i] != ';') { p.fail("';' expected"); return IPKGPU_ERR_INVALID; }
    t->root = root;
    t->index();
    *out = t.release();
    return IPKGPU_OK;
}

int ipkgpu_tree_load(const char* path, ipkgpu_tree** out)
{
    if (!path || !out) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    FILE* f = fopen(path, "rb");
    if (!f) { g_tree_err = std::string("cannot open ") + path; return IPKGPU_ERR_INVALID; }
    std::string text;
    char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    fclose(f);
    return ipkgpu_tree_parse(text.c_str(), out);
}

void ipkgpu_tree_free(ipkgpu_tree* t) { delete t; }
uint32_t ipkgpu_tree_num_nodes(const ipkgpu_tree* t) { return t ? (uint32_t)t->post.size() : 0; }
uint32_t ipkgpu_tree_num_leaves(const ipkgpu_tree* t) { return t && t->root >= 0 ? t->nodes[t->root].num_leaves : 0; }
int ipkgpu_tree_is_rooted(const ipkgpu_tree* t) { return t && t->is_rooted() ? 1 : 0; }
const char* ipkgpu_tree_label(const ipkgpu_tree* t, uint32_t postorder_id)
{
    return t && postorder_id < t->post.size() ? t->nodes[t->post[postorder_id]].label.c_str() : nullptr;
}
int64_t ipkgpu_tree_parent(const ipkgpu_tree* t, uint32_t postorder_id)
{
    if (!t || postorder_id >= t->post.size()) return -1;
    const int p = t->nodes[t->post[postorder_id]].parent;
    return p < 0 ? -1 : (int64_t)t->nodes[p].postorder;
}
double ipkgpu_tree_branch_length(const ipkgpu_tree* t, uint32_t postorder_id)
{
    return t && postorder_id < t->post.size() ? t->nodes[t->post[postorder_id]].length : 0.0;
}
const char* ipkgpu_tree_newick(ipkgpu_tree* t)
{
    if (!t || t->root < 0) return nullptr;
    t->newick_cache.clear();
    write_newick(*t, t->root, t->newick_cache);
    t->newick_cache.push_back(';');
    return t->newick_cache.c_str();
}

// (num_nodes, subtree_branch_length) per node in post-order: the database header's tree index (db_builder.cpp:192-197;
// visit_subtree's default iterator is the post-order one, as in ar.cpp:803-806)
int ipkgpu_tree_index(const ipkgpu_tree* t, uint32_t* num_nodes, double* subtree_length)
{
    if (!t || !num_nodes || !subtree_length) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    for (size_t i = 0; i < t->post.size(); ++i) {
        num_nodes[i] = t->nodes[t->post[i]].num_nodes;
        subtree_length[i] = t->nodes[t->post[i]].subtree_length;
    }
    return IPKGPU_OK;
}

// tree_extender::extend (extended_tree.cpp:76-150)
int ipkgpu_tree_extend(const ipkgpu_tree* original, ipkgpu_tree** out)
{
    if (!original || !out) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ipkgpu_tree> ext(clone(*original));
    size_t counter = original->post.size() + 1;                              // :80
    // extend_subtree recurses children-first over a COPY of each child list, i.e. visits the original nodes in post-order;
    // the arena indices of the copy equal the original's, whose post-order ids are still the old ones (:118-121)
    for (int id : original->post) {
        const int parent = ext->nodes[id].parent;
        if (parent < 0) continue;                                            // :110 (root)
        const Node& orig = original->nodes[id];
        // calc_ghost_branch_lengths (:36-73)
        const double old_len = orig.length;
        const double x0_len = old_len / 2.0;
        const double residual = old_len - x0_len;
        double x1_len;
        if (orig.children.empty()) x1_len = residual;
        else x1_len = (total_branch_length(*original, id) + residual * orig.num_leaves) / orig.num_leaves;
        const std::string x0_name = std::to_string(counter++) + "_X0";
        // parent->remove_child(node); parent->add_child(x0)   (:126-127: x0 goes to the END of the parent's child list)
        ext->remove_child(parent, id);
        const int x0 = ext->add(x0_name, x0_len, true, parent);
        const std::string x1_name = std::to_string(counter++) + "_X1";
        const int x1 = ext->add(x1_name, x1_len, true, x0);                  // x0's children: x1, then the node (:132-133)
        ext->nodes[x0].children.push_back(id);
        ext->nodes[id].parent = x0;
        ext->nodes[id].length = ext->nodes[id].length - x0_len;              // :134-135
        ext->add(std::to_string(counter++) + "_X2", 0.01, true, x1);         // :137-140
        ext->add(std::to_string(counter++) + "_X3", 0.01, true, x1);
        ext->ghost_to_branch[x0_name] = orig.postorder;                      // :145-146
        ext->ghost_to_branch[x1_name] = orig.postorder;
    }
    ext->index();
    *out = ext.release();
    return IPKGPU_OK;
}

// reroot_tree (extended_tree.cpp:186-205): a root with more than two children (a, b, c); becomes ((b, c), a)added_root;
int ipkgpu_tree_reroot(ipkgpu_tree* t)
{
    if (!t || t->root < 0) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    const int root = t->root;
    if (t->nodes[root].children.size() > 2) {
        const int a = t->nodes[root].children[0];
        const int nr = t->add("added_root", 0.0, false, -1);
        t->nodes[nr].children.push_back(root); t->nodes[root].parent = nr;   // add_child(root), add_child(a)
        t->nodes[nr].children.push_back(a);
        t->remove_child(root, a);
        t->nodes[a].parent = nr;
        t->root = nr;
        t->index();
    }
    return IPKGPU_OK;
}

// get_ghost_ids + group_ghost_ids + map_nodes + get_submatrices' label lookup, in one plan:
// for every ghost node kept by the strategy, in the order explore_kmers scores them (groups in first-seen order, a
// group's ghosts in tree order): its extended-tree label, the AR tree's label of the same node, the branch id.
//   strategy: 0 = both, 1 = inner only (_X0), 2 = outer only (_X1)      (db_builder.cpp:495-507)
int ipkgpu_ghost_plan_make(const ipkgpu_tree* original, const ipkgpu_tree* extended, const ipkgpu_tree* ar_tree, int strategy,
                           ipkgpu_ghost_plan** out)
{
    if (!original || !extended || !out) { g_tree_err = "null argument"; return IPKGPU_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ipkgpu_ghost_plan> plan(new ipkgpu_ghost_plan());
    // map_nodes (ar.cpp:790-834): both trees in post-order, node by node; unlabelled extended nodes are skipped
    std::unordered_map<std::string, std::string> ext_to_ar;
    if (ar_tree) {
        if (extended->post.size() != ar_tree->post.size()) {
            g_tree_err = "extended tree and AR tree differ in the number of nodes: " + std::to_string(extended->post.size()) +
                         " vs. " + std::to_string(ar_tree->post.size());
            return IPKGPU_ERR_INVALID;
        }
        for (size_t i = 0; i < extended->post.size(); ++i) {
            const std::string& lab = extended->nodes[extended->post[i]].label;
            if (!lab.empty()) ext_to_ar[lab] = ar_tree->nodes[ar_tree->post[i]].label;
        }
    }
    // get_ghost_ids: tree iteration order (post-order); group_ghost_ids: first-seen groups, root's ghosts skipped
    std::vector<std::vector<std::string>> groups;
    std::vector<uint32_t> group_branch;
    std::unordered_map<uint32_t, size_t> index_of;
    const uint32_t orig_root = original->nodes[original->root].postorder;
    for (int id : extended->post) {
        const std::string& lab = extended->nodes[id].label;
        const bool x0 = ends_with(lab, "_X0"), x1 = ends_with(lab, "_X1");
        if (!(strategy == 1 ? x0 : strategy == 2 ? x1 : (x0 || x1))) continue;
        auto it = extended->ghost_to_branch.find(lab);
        if (it == extended->ghost_to_branch.end()) { g_tree_err = "ghost node " + lab + " has no branch (tree not produced by ipkgpu_tree_extend)"; return IPKGPU_ERR_INVALID; }
        const uint32_t branch = it->second;
        if (branch == orig_root) continue;
        auto g = index_of.find(branch);
        if (g == index_of.end()) { index_of[branch] = groups.size(); groups.push_back({lab}); group_branch.push_back(branch); }
        else groups[g->second].push_back(lab);
    }
    for (size_t g = 0; g < groups.size(); ++g)
        for (const std::string& lab : groups[g]) {
            plan->ext_labels.push_back(lab);
            if (ar_tree) {
                auto it = ext_to_ar.find(lab);
                if (it == ext_to_ar.end()) { g_tree_err = "no AR node for " + lab; return IPKGPU_ERR_INVALID; }
                plan->ar_labels.push_back(it->second);
            } else plan->ar_labels.push_back(lab);
            plan->branches.push_back(group_branch[g]);
        }
    plan->tree_num_nodes.resize(original->post.size());
    plan->tree_subtree_length.resize(original->post.size());
    ipkgpu_tree_index(original, plan->tree_num_nodes.data(), plan->tree_subtree_length.data());
    *out = plan.release();
    return IPKGPU_OK;
}
void ipkgpu_ghost_plan_free(ipkgpu_ghost_plan* p) { delete p; }
uint32_t ipkgpu_ghost_plan_size(const ipkgpu_ghost_plan* p) { return p ? (uint32_t)p->branches.size() : 0; }
const char* ipkgpu_ghost_plan_ext_label(const ipkgpu_ghost_plan* p, uint32_t i) { return p && i < p->ext_labels.size() ? p->ext_labels[i].c_str() : nullptr; }
const char* ipkgpu_ghost_plan_ar_label(const ipkgpu_ghost_plan* p, uint32_t i) { return p && i < p->ar_labels.size() ? p->ar_labels[i].c_str() : nullptr; }
const uint32_t* ipkgpu_ghost_plan_branches(const ipkgpu_ghost_plan* p) { return p ? p->branches.data() : nullptr; }

}  // extern "C"
