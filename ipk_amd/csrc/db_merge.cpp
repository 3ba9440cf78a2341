// db_merge.cpp -- the database file of a multi-GPU build: streaming P-way merge of the ranks' shard files (host code).
//
// Reference: the on-disk path's final stage, merge_stage2 (ipk/src/db_builder.cpp:392-458) -- every batch file is opened with
// a batch_loader, a priority queue ordered by batch_loader_compare (filter value) hands out the k-mer to write next, and the
// output archive is appended to k-mer by k-mer (:444); memory is one k-mer per open batch.  Here the "batches" are the ranks'
// shards: rank r owns the k-mers with code % P == r, computes their filter values itself (they are per k-mer) and writes them,
// already in ITS filter order, as a database file of its own (ipkgpu_db_write / ipkgpu_db_write_host).  This merge reads the P
// shards through bounded buffers, always copies the record with the smallest (filter value, key) next, and writes the header
// with the summed totals first -- the file one GPU writes for the same input, byte for byte, with a resident set that does not
// depend on the number of entries.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ipkgpu.h"
#include "ipk_format.hpp"

namespace {

thread_local std::string g_merge_err;

struct ShardReader {
    FILE* f = nullptr;
    std::vector<uint8_t> buf;          // [lo, hi) = bytes read and not yet consumed
    size_t lo = 0, hi = 0;
    uint64_t left = 0;                 // records not yet handed out
    uint64_t key = 0;                  // sort key of the current record
    uint64_t rec_bytes = 0;            // its size
    ~ShardReader() { if (f) fclose(f); }

    // at least n unconsumed bytes in the buffer (false: the file ends first)
    bool need(size_t n)
    {
        if (hi - lo >= n) return true;
        if (lo) { memmove(buf.data(), buf.data() + lo, hi - lo); hi -= lo; lo = 0; }
        if (buf.size() < n) buf.resize(n);
        while (hi < n) {
            const size_t got = fread(buf.data() + hi, 1, buf.size() - hi, f);
            if (got == 0) return false;
            hi += got;
        }
        return true;
    }
    // loads the head of the next record; false on a truncated file
    bool next()
    {
        if (left == 0) return true;
        if (!need(ipkfmt::RECORD_HEAD_BYTES)) return false;
        uint32_t w[4];
        memcpy(w, buf.data() + lo, sizeof w);
        const uint64_t n = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        rec_bytes = ipkfmt::record_bytes(n);
        key = ipkfmt::record_sort_key(w[0], w[1]);
        return need((size_t)rec_bytes);
    }
};

}  // namespace

extern "C" {

const char* ipkgpu_db_merge_last_error(void) { return g_merge_err.c_str(); }
uint32_t ipkgpu_db_protocol_version(void) { return ipkfmt::protocol_version(); }

int ipkgpu_db_merge_files(const ipkgpu_db_header* h, const char* const* shard_paths, uint32_t n_shards, const char* path,
                          uint64_t* total_kmers, uint64_t* total_entries, uint64_t* bytes_written)
{
    if (!h || !path || (n_shards && !shard_paths)) { g_merge_err = "null argument"; return IPKGPU_ERR_INVALID; }
    constexpr size_t IN_BUF = (size_t)4 << 20, OUT_BUF = (size_t)16 << 20;
    std::vector<std::unique_ptr<ShardReader>> in;
    uint64_t nk = 0, ne = 0;
    for (uint32_t s = 0; s < n_shards; ++s) {
        std::unique_ptr<ShardReader> r(new ShardReader());
        r->f = fopen(shard_paths[s], "rb");
        if (!r->f) { g_merge_err = std::string("cannot open shard ") + shard_paths[s]; return IPKGPU_ERR_INVALID; }
        setvbuf(r->f, nullptr, _IONBF, 0);                             // (own buffering: no second copy through stdio)
        uint64_t k = 0, e = 0;
        if (!ipkfmt::read_head(r->f, k, e)) { g_merge_err = std::string("not a database shard: ") + shard_paths[s]; return IPKGPU_ERR_INVALID; }
        r->left = k; nk += k; ne += e;
        r->buf.resize(IN_BUF);
        if (!r->next()) { g_merge_err = std::string("truncated shard: ") + shard_paths[s]; return IPKGPU_ERR_INVALID; }
        in.push_back(std::move(r));
    }
    FILE* out = fopen(path, "wb");
    if (!out) { g_merge_err = std::string("cannot create ") + path; return IPKGPU_ERR_INVALID; }
    struct Close { FILE*& f; ~Close() { if (f) fclose(f); } } closer{out};
    setvbuf(out, nullptr, _IONBF, 0);
    const std::vector<uint8_t> head = ipkfmt::file_head(h->sequence_type, h->tree_index_size, h->tree_num_nodes, h->tree_subtree_length,
                                                        h->newick, h->kmer_size, h->omega, nk, ne);
    uint64_t total = 0;
    std::vector<uint8_t> ob;
    ob.reserve(OUT_BUF + ((size_t)1 << 20));
    auto flush = [&]() -> bool {
        if (ob.empty()) return true;
        const bool ok = fwrite(ob.data(), 1, ob.size(), out) == ob.size();
        total += ob.size(); ob.clear();
        return ok;
    };
    ob.insert(ob.end(), head.begin(), head.end());
    for (uint64_t done = 0; done < nk; ++done) {
        // the shard whose current record comes first (P is the number of GPUs: a scan beats a heap)
        ShardReader* best = nullptr;
        for (auto& r : in) if (r->left && (!best || r->key < best->key)) best = r.get();
        if (!best) { g_merge_err = "shard totals and records disagree"; return IPKGPU_ERR_INVALID; }
        ob.insert(ob.end(), best->buf.data() + best->lo, best->buf.data() + best->lo + best->rec_bytes);
        best->lo += (size_t)best->rec_bytes;
        best->left -= 1;
        if (!best->next()) { g_merge_err = "truncated shard"; return IPKGPU_ERR_INVALID; }
        if (ob.size() >= OUT_BUF && !flush()) { g_merge_err = "write failed"; return IPKGPU_ERR_INVALID; }
    }
    if (!flush()) { g_merge_err = "write failed"; return IPKGPU_ERR_INVALID; }
    FILE* f = out; out = nullptr;
    if (fclose(f) != 0) { g_merge_err = "close failed"; return IPKGPU_ERR_INVALID; }
    if (total_kmers) *total_kmers = nk;
    if (total_entries) *total_entries = ne;
    if (bytes_written) *bytes_written = total;
    return IPKGPU_OK;
}

}  // extern "C"
