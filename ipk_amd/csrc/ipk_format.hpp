// ipk_format.hpp -- the ONE place that knows the bytes of the database file.
//
// IPK streams its database through i2l::save_header / i2l::save_phylo_kmer over a boost::archive::binary_oarchive opened on
// the output file (ipk/src/db_builder.cpp:145-146,176-177,297-306,323-327).  i2l and Boost are not in the reference tree,
// so this layout is a RECONSTRUCTION -- unpinned: no real .ipk file or i2l source was available to check it against.  It
// follows what the call sites fix (field order of ipk_header at db_builder.cpp:297-305; per k-mer: key, filter value,
// entries of (branch, score) at :323-327) and writes every primitive the way Boost's binary_oarchive does on x86-64
// (native little endian; std::string = u64 length + bytes; size_t = u64).  A maintainer with i2l at hand changes only
// this header.
//
//   preamble  u64 22, "serialization::archive", u16 library version, u8 sizeof(int), u8 sizeof(long), u8 sizeof(float),
//             u8 sizeof(double), i32 1          (basic_binary_oarchive::init + basic_binary_oprimitive::init)
//   protocol  u32 protocol version                                            -- GUESS, see below; absent when the version is 0
//   header    string sequence_type ; u8 positions_loaded (bool; only with the protocol word) ;
//             u64 n ; n x { u64 num_nodes ; f64 subtree_branch_length } ; string newick ;
//             u64 kmer_size ; f32 omega ; u64 total_num_kmers ; u64 total_num_entries
//   k-mers    per k-mer in filter order: u32 key ; f32 filter_value ; u64 n ; n x { u32 branch ; f32 score }
//             positioned build (ipk-aa-pos, KEEP_POSITIONS: db_builder.cpp:655-662,687-689; positions flag = 1):
//             n x { u32 branch ; f32 score ; u16 position } -- the position's width is a GUESS too (i2l's pos_type is un-vendored)
//
// Protocol version and positions flag.  A LOADED database answers version() and positions_loaded() (the reference's own
// ipkdiff compares both: tools/src/diff.cpp:41-46,137-145; the positions check is commented out there as "broken in v0.4.x+"),
// and neither is an argument of phylo_kmer_db's constructor (db_builder.cpp:174) -- so both are read from the file, which the
// seven fields of ipk_header (db_builder.cpp:297-305) alone cannot carry.  WHERE they sit and WHAT number the current protocol has
// are not visible from the reference tree: here the version is the first word behind the Boost preamble (a reader must know it
// before it can parse anything else) and the flag follows the sequence type (the order in which ipkdiff reports them).  Which
// fields are guesses: the protocol word's position, width (u32) and value; the flag's position; and, as before, the archive
// library version, u64/f64 tree index, f32 filter value and u32 key widths.  IPKGPU_IPK_PROTOCOL_VERSION=0 leaves both fields out
// (the round-3 layout); any other value is written as given.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define IPKFMT_HD __host__ __device__
#else
#define IPKFMT_HD
#endif

namespace ipkfmt {

constexpr uint16_t BOOST_ARCHIVE_LIBRARY_VERSION = 19;      // Boost 1.74 .. 1.83 (ASSUMPTION: the build's Boost is unknown)
// the version actually written: IPKGPU_BOOST_ARCHIVE_VERSION in the environment overrides the assumption (a reader built against
// another Boost rejects an archive whose library version is newer than its own)
inline uint16_t archive_library_version()
{
    if (const char* e = getenv("IPKGPU_BOOST_ARCHIVE_VERSION")) { const long v = atol(e); if (v > 0 && v < 65536) return (uint16_t)v; }
    return BOOST_ARCHIVE_LIBRARY_VERSION;
}
// CHANGELOG.txt counts the protocol changes (v0.2.0: three versions 2..4 by then; v0.3.0; v0.4.0; v0.5.0 "sorted by MI") -> 7
constexpr uint32_t IPK_PROTOCOL_VERSION = 7;                // ASSUMPTION (i2l/version.h is un-vendored)
inline uint32_t protocol_version()
{
    if (const char* e = getenv("IPKGPU_IPK_PROTOCOL_VERSION")) { const long v = atol(e); if (v >= 0 && v < (1l << 31)) return (uint32_t)v; }
    return IPK_PROTOCOL_VERSION;
}
constexpr uint64_t RECORD_HEAD_BYTES = 16;                   // key, filter value, entry count
constexpr uint64_t ENTRY_BYTES = 8;                          // branch, score
constexpr uint64_t ENTRY_POS_BYTES = 10;                     // branch, score, position (u16: ASSUMPTION)

IPKFMT_HD inline uint64_t record_bytes(uint64_t n_entries) { return RECORD_HEAD_BYTES + ENTRY_BYTES * n_entries; }

// the four 32-bit words of a record's head
IPKFMT_HD inline void record_head(uint32_t key, uint32_t fv_bits, uint64_t n_entries, uint32_t (&w)[4])
{
    w[0] = key; w[1] = fv_bits; w[2] = (uint32_t)n_entries; w[3] = (uint32_t)(n_entries >> 32);
}

inline void put(std::vector<uint8_t>& o, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); }
template <class T> inline void put_v(std::vector<uint8_t>& o, T v) { put(o, &v, sizeof v); }
inline void put_string(std::vector<uint8_t>& o, const char* s) { const uint64_t n = s ? strlen(s) : 0; put_v<uint64_t>(o, n); put(o, s, n); }

// everything in front of the first k-mer record
inline std::vector<uint8_t> file_head(const char* sequence_type, uint64_t n_index, const uint32_t* num_nodes, const double* subtree_length,
                                      const char* newick, uint64_t kmer_size, float omega, uint64_t total_kmers, uint64_t total_entries,
                                      bool positions_loaded = false)
{
    std::vector<uint8_t> o;
    put_string(o, "serialization::archive");
    put_v<uint16_t>(o, archive_library_version());
    put_v<uint8_t>(o, 4); put_v<uint8_t>(o, 8); put_v<uint8_t>(o, 4); put_v<uint8_t>(o, 8);
    put_v<int32_t>(o, 1);
    const uint32_t proto = protocol_version();
    if (proto) put_v<uint32_t>(o, proto);
    put_string(o, sequence_type);
    if (proto) put_v<uint8_t>(o, positions_loaded ? 1 : 0);
    put_v<uint64_t>(o, n_index);
    for (uint64_t i = 0; i < n_index; ++i) { put_v<uint64_t>(o, num_nodes[i]); put_v<double>(o, subtree_length[i]); }
    put_string(o, newick);
    put_v<uint64_t>(o, kmer_size);
    put_v<float>(o, omega);
    put_v<uint64_t>(o, total_kmers);
    put_v<uint64_t>(o, total_entries);
    return o;
}

// ---- reading back what file_head wrote (the shard files of a multi-GPU build are database files themselves) ----------------
inline bool get(FILE* f, void* p, size_t n) { return n == 0 || fread(p, 1, n, f) == n; }
template <class T> inline bool get_v(FILE* f, T& v) { return get(f, &v, sizeof v); }
inline bool skip_string(FILE* f, uint64_t limit = (uint64_t)1 << 32)
{
    uint64_t n = 0;
    return get_v(f, n) && n <= limit && fseek(f, (long)n, SEEK_CUR) == 0;
}
// Positions `f` at the first k-mer record; the header's totals come back.  false: not a file of this layout (as written by
// this process: the protocol word is expected exactly when protocol_version() is non-zero, and must hold that value).
inline bool read_head(FILE* f, uint64_t& total_kmers, uint64_t& total_entries)
{
    uint64_t n = 0;
    char magic[22];
    if (!get_v(f, n) || n != 22 || !get(f, magic, 22) || memcmp(magic, "serialization::archive", 22) != 0) return false;
    uint16_t ver; uint8_t sz[4]; int32_t one;
    if (!get_v(f, ver) || !get(f, sz, 4) || !get_v(f, one) || sz[0] != 4 || sz[1] != 8 || sz[2] != 4 || sz[3] != 8 || one != 1) return false;
    uint64_t n_index = 0, kmer_size = 0; float omega = 0;
    const uint32_t proto = protocol_version();
    uint32_t got = 0; uint8_t positions = 0;
    if (proto && (!get_v(f, got) || got != proto)) return false;
    if (!skip_string(f) || (proto && (!get_v(f, positions) || positions > 1)) || !get_v(f, n_index) || n_index > ((uint64_t)1 << 32) || fseek(f, (long)(n_index * 16), SEEK_CUR) != 0) return false;
    if (!skip_string(f) || !get_v(f, kmer_size) || !get_v(f, omega)) return false;
    return get_v(f, total_kmers) && get_v(f, total_entries);
}
// the order of the k-mer records: ascending filter value (as an order-preserving integer code of the float), ties by
// ascending key -- kernels_filter.hpp's sort key, db_builder.cpp:281-284 (`std::sort` over kmer_fv)
inline uint64_t record_sort_key(uint32_t key, uint32_t fv_bits)
{
    const uint32_t code = (fv_bits & 0x80000000u) ? ~fv_bits : (fv_bits | 0x80000000u);
    return ((uint64_t)code << 32) | key;
}

}  // namespace ipkfmt
