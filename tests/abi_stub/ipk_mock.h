// ipk_mock.h -- the sliver of IPK / i2l that INTEGRATION.md's db_builder patch touches, so that the patch can be compiled
// against include/ipkgpu.h without the (un-vendored) i2l library.  TEST INFRASTRUCTURE: names and member signatures follow
// ipk/src/db_builder.cpp:115-170,495-627 and ipk/include/window.h:20-72; nothing here is shipped.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

struct seq_traits { static constexpr size_t alphabet_size = 4; };
namespace phylo_kmer_ns { using branch_type = uint32_t; using key_type = uint32_t; using score_type = float; }
struct phylo_kmer { using branch_type = phylo_kmer_ns::branch_type; using key_type = phylo_kmer_ns::key_type; using score_type = phylo_kmer_ns::score_type; };
using branch_type = phylo_kmer::branch_type;
inline float score_threshold(float omega, size_t k) { return std::pow(omega / (float)seq_traits::alphabet_size, (float)k); }

struct pkdb_value { branch_type branch; float score; };
struct phylo_kmer_db {
    std::unordered_map<uint32_t, std::vector<pkdb_value>> map;
    void unsafe_insert(uint32_t key, const pkdb_value& v) { map[key].push_back(v); }
};

namespace ipk {
struct matrix {                                           // window.h:20-72
    using column = std::array<float, seq_traits::alphabet_size>;
    std::vector<column> _data;
    const std::vector<column>& get_data() const { return _data; }
    void clear() { _data.clear(); }
};
enum class ghost_strategy { BOTH };
}
struct phylo_tree {};

class db_builder {
public:
    using id_group = std::vector<std::string>;
    using proba_group = std::vector<std::reference_wrapper<ipk::matrix>>;
    std::tuple<std::vector<phylo_kmer::branch_type>, size_t> explore_kmers();
    // the members the patch reads (db_builder.cpp:115-146)
    phylo_tree _extended_tree;
    ipk::ghost_strategy _ghost_strategy = ipk::ghost_strategy::BOTH;
    std::unordered_map<std::string, branch_type> _extended_mapping;
    std::unordered_map<std::string, ipk::matrix> _matrices;
    size_t _kmer_size = 8;
    float _omega = 1.5f;
    phylo_kmer_db _phylo_kmer_db;
    std::vector<id_group> _groups;
    std::vector<id_group> group_ghost_ids(const std::vector<std::string>&) const { return _groups; }
    proba_group get_submatrices(const id_group& g) { proba_group r; for (auto& l : g) r.push_back(std::ref(_matrices.at(l))); return r; }
};
inline std::vector<std::string> get_ghost_ids(const phylo_tree&, ipk::ghost_strategy) { return {}; }
