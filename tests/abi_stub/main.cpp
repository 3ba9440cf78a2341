// Drives the INTEGRATION.md patch (spliced in by tests/test_abi_compile.py as integration_stub.inc) on a tiny synthetic input.
// Exit codes: 0 = ran on a GPU and produced k-mers, 3 = no GPU (the library refused: it has no CPU fallback), 1 = failure.
// argv[1] (optional): file that receives the input matrices (raw float32 [6][60][4]) so that the test can run the oracle on
// the very same numbers; the printed line carries the threshold's bits, the scored count and an FNV-1a of the database
// (every (key, branch, score bits) triple, ordered by key then branch).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ipk_mock.h"
#include "integration_stub.inc"
#include "integration_multi.inc"

int main(int argc, char** argv)
{
    db_builder b;
    std::vector<float> dump;
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
                dump.insert(dump.end(), p.begin(), p.end());
            }
            b._matrices.emplace(label, std::move(m));
            b._extended_mapping[label] = (branch_type)(7 + g);
            grp.push_back(label);
        }
        b._groups.push_back(grp);
    }
    if (argc > 1) {
        FILE* f = std::fopen(argv[1], "wb");
        if (!f || std::fwrite(dump.data(), sizeof(float), dump.size(), f) != dump.size()) { std::fprintf(stderr, "cannot write %s\n", argv[1]); return 1; }
        std::fclose(f);
    }
    try {
        auto [ids, count] = b.explore_kmers();
        std::vector<std::array<uint32_t, 3>> all;
        for (const auto& [key, entries] : b._phylo_kmer_db.map)
            for (const auto& e : entries) { uint32_t bits; std::memcpy(&bits, &e.score, 4); all.push_back({key, e.branch, bits}); }
        std::sort(all.begin(), all.end());
        uint64_t fnv = 1469598103934665603ull;
        for (const auto& t : all)
            for (uint32_t w : t)
                for (int i = 0; i < 4; ++i) { fnv ^= (w >> (8 * i)) & 0xFFu; fnv *= 1099511628211ull; }
        const float log_eps = std::log10(score_threshold(b._omega, b._kmer_size));          // as the patch computes it (db_builder.cpp:640)
        uint32_t eps_bits; std::memcpy(&eps_bits, &log_eps, 4);
        std::printf("groups %zu scored %zu kmers %zu entries %zu eps %08x fnv %016llx\n", ids.size(), count, b._phylo_kmer_db.map.size(),
                    all.size(), eps_bits, (unsigned long long)fnv);
        return (ids.size() == 3 && count > 0 && !b._phylo_kmer_db.map.empty()) ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "explore_kmers: %s\n", e.what());
        return std::string(e.what()).find("no CPU fallback") != std::string::npos ? 3 : 1;
    }
}
