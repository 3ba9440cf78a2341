// Drives the INTEGRATION.md patch (spliced in by tests/test_abi_compile.py as integration_stub.inc) on a tiny synthetic input.
// Exit codes: 0 = ran on a GPU and produced k-mers, 3 = no GPU (the library refused: it has no CPU fallback), 1 = failure.
#include <cstdio>
#include <cstdlib>
#include "ipk_mock.h"
#include "integration_stub.inc"
#include "integration_multi.inc"

int main()
{
    db_builder b;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f; };
    for (int g = 0; g < 3; ++g) {
        db_builder::id_group grp;
        for (int x = 0; x < 2; ++x) {
            const std::string label = std::to_string(100 + 2 * g + x) + (x ? "_X1" : "_X0");
            ipk::matrix m;
            for (int site = 0; site < 60; ++site) {
                std::array<float, 4> p;
                float sum = 0;
                for (auto& v : p) { v = std::pow(rnd(), 8.0f) + 1e-6f; sum += v; }
                for (auto& v : p) v = std::log10(v / sum);
                m._data.push_back(p);
            }
            b._matrices.emplace(label, std::move(m));
            b._extended_mapping[label] = (branch_type)(7 + g);
            grp.push_back(label);
        }
        b._groups.push_back(grp);
    }
    try {
        auto [ids, count] = b.explore_kmers();
        std::printf("groups %zu scored %zu kmers %zu\n", ids.size(), count, b._phylo_kmer_db.map.size());
        return (ids.size() == 3 && count > 0 && !b._phylo_kmer_db.map.empty()) ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "explore_kmers: %s\n", e.what());
        return std::string(e.what()).find("no CPU fallback") != std::string::npos ? 3 : 1;
    }
}
