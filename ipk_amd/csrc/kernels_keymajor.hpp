// kernels_keymajor.hpp -- dense per-group tables -> key-major database parts, and their merge.
//
// Device analogue of the tail of explore_group (ipk/src/db_builder.cpp:685-694:
// `_phylo_kmer_db.unsafe_insert(kmer, {branch, score})` group after group) and of the on-disk path's
// k-mer-keyed partition + merge (branch_group.cpp:45-70,104-107; db_builder.cpp:340-458):
//
//   part   for owner o of P: for every k-mer code x with x % P == o (ascending), the entries
//          (branch id, score) of the groups that scored x, in group order.
//          counts[o][x / P] = number of entries; entries laid out owner-major, then key, then group.
//   merge  concatenates, per key, the entries of S sources (ranks in rank order, or batches in batch
//          order) -- i.e. global group order, the order the reference appends them.
//
// Everything is a coalesced sweep of the dense tables: no sort, no global atomics.
#pragma once
#include <type_traits>
#include "dcla_device.hpp"
#include "comp_table.hpp"

namespace ipkgpu {

typedef const uint32_t __attribute__((address_space(1)))* global_u32_ptr;      // a pointer the compiler knows to be global memory (no flat_load)


// counts_perm[(x % P) * slots + x / P] = #groups g in [0, G) with table[g][x] != 0
__global__ __launch_bounds__(256) void km_count_kernel(const uint32_t* __restrict__ table, uint64_t T, uint32_t G,
                                                       uint32_t P, uint64_t slots, uint32_t* __restrict__ counts)
{
    const uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= T) return;
    uint32_t c = 0;
    const uint32_t* t = table + x;
    uint32_t g = 0;
    for (; g + 4 <= G; g += 4) {
        const uint32_t v0 = t[(size_t)(g + 0) * T], v1 = t[(size_t)(g + 1) * T];
        const uint32_t v2 = t[(size_t)(g + 2) * T], v3 = t[(size_t)(g + 3) * T];
        c += (v0 != 0u) + (v1 != 0u) + (v2 != 0u) + (v3 != 0u);
    }
    for (; g < G; ++g) c += (t[(size_t)g * T] != 0u);
    const uint64_t o = x % P, q = x / P;
    counts[o * slots + q] = c;               // padded slots (q * P + o >= T) stay at their memset 0
}

// The same counts from the occupancy bits the LDS reduce passes leave behind (kernels_score.hpp,
// store_slice_mask): 1/32 of the bytes of the dense tables.  One thread per mask word (32 keys); the words of
// successive groups are added into eight bit planes (a ripple-carry add of a 1-bit number: 16 logic operations for
// 32 keys), emptied into the 32 per-key totals every 255 groups.
__global__ __launch_bounds__(256) void km_count_mask_kernel(const uint32_t* __restrict__ mask, uint64_t W, uint64_t T,
                                                            uint32_t G, uint32_t P, uint64_t slots, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ qpack)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w * 32 >= T) return;
    uint32_t plane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t total[32], pack[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) { total[b] = 0; pack[b] = 0; }
    const uint32_t* m = mask + w;
    // qpack (G <= 256): the rows are also counted per quarter of ceil(G / 4) rows -- byte q of qpack[key] = rows of quarter q
    // that hold the key (what km_write_c_kernel's four wavefronts start from); otherwise chunks of 255 rows (8 bit planes)
    const uint32_t step = qpack ? (G + 3) / 4 : 255u;
    uint32_t q = 0;
    for (uint32_t g0 = 0; g0 < G; g0 += step, ++q) {
        const uint32_t g1 = min(G, g0 + step);
        uint32_t g = g0;
        for (; g + 8 <= g1; g += 8) {                      // eight rows' loads in flight together
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = m[(size_t)(g + u) * W];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t carry = v[u];
#pragma unroll
                for (int p = 0; p < 8; ++p) { const uint32_t t = plane[p] & carry; plane[p] ^= carry; carry = t; }
            }
        }
        for (; g < g1; ++g) {
            uint32_t carry = m[(size_t)g * W];
#pragma unroll
            for (int p = 0; p < 8; ++p) { const uint32_t t = plane[p] & carry; plane[p] ^= carry; carry = t; }
        }
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            uint32_t c = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) c |= ((plane[p] >> b) & 1u) << p;
            total[b] += c;
            pack[b] |= c << ((8u * q) & 31u);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) plane[p] = 0;
    }
    if (P == 1 && w * 32 + 32 <= T) {
        // one owner: the word's 32 counts are consecutive -- eight 16-byte stores per array instead of thirty-two 4-byte ones a line apart
        uint4* c4 = reinterpret_cast<uint4*>(counts + w * 32);
        uint4* q4 = reinterpret_cast<uint4*>(qpack + w * 32);
#pragma unroll
        for (int b = 0; b < 32; b += 4) {
            c4[b >> 2] = make_uint4(total[b], total[b + 1], total[b + 2], total[b + 3]);
            if (qpack) q4[b >> 2] = make_uint4(pack[b], pack[b + 1], pack[b + 2], pack[b + 3]);
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        const uint64_t x = w * 32 + b;
        if (x < T) {
            const uint64_t idx = (x % P) * slots + x / P;
            counts[idx] = total[b];
            if (qpack) qpack[idx] = pack[b];
        }
    }
}

// One thread per key: for key spaces too small to fill the chip with one thread per mask word (DNA k <= 11).
__global__ __launch_bounds__(256) void km_count_mask_key_kernel(const uint32_t* __restrict__ mask, uint64_t W, uint64_t T,
                                                                uint32_t G, uint32_t P, uint64_t slots, uint32_t* __restrict__ counts,
                                                                uint32_t* __restrict__ qpack)
{
    const uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= T) return;
    const uint32_t* m = mask + (x >> 5);
    const uint32_t sh = (uint32_t)x & 31u;
    const uint32_t step = qpack ? (G + 3) / 4 : G;          // (per-quarter counts: see km_count_mask_kernel)
    uint32_t total = 0, pack = 0, q = 0;
    for (uint32_t g0 = 0; g0 < G; g0 += step, ++q) {
        const uint32_t g1 = min(G, g0 + step);
        uint32_t c = 0, g = g0;
        for (; g + 16 <= g1; g += 16) {                     // sixteen rows' loads in flight together
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = m[(size_t)(g + u) * W];
#pragma unroll
            for (int u = 0; u < 16; ++u) c += (v[u] >> sh) & 1u;
        }
        for (; g < g1; ++g) c += (m[(size_t)g * W] >> sh) & 1u;
        total += c;
        pack |= c << ((8u * q) & 31u);
    }
    const uint64_t idx = (x % P) * slots + x / P;
    counts[idx] = total;
    if (qpack) qpack[idx] = pack;
}

// ---- generic exclusive scan of u32 -> u64 (three kernels) -------------------------------------
constexpr uint32_t SCAN_BLOCK = 4096;

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                              uint32_t* __restrict__ sums)
{
    __shared__ uint32_t wsum[4];
    const uint64_t s0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
    const uint32_t m = (uint32_t)min((uint64_t)SCAN_BLOCK, n - s0);
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < m; i += 256) acc += in[s0 + i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// out[i] = block_off[i / SCAN_BLOCK] + sum(in[block start .. i)); thread t owns 16 consecutive items
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                         const uint64_t* __restrict__ block_off,
                                                         uint64_t* __restrict__ out)
{
    __shared__ uint32_t tsum[256];
    const uint64_t s0 = (uint64_t)blockIdx.x * SCAN_BLOCK;
    const uint32_t m = (uint32_t)min((uint64_t)SCAN_BLOCK, n - s0);
    constexpr uint32_t PER = SCAN_BLOCK / 256;
    const uint32_t lo = threadIdx.x * PER;
    uint32_t v[PER];
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) { v[i] = (lo + i < m) ? in[s0 + lo + i] : 0u; acc += v[i]; }
    tsum[threadIdx.x] = acc;
    __syncthreads();
    // exclusive scan of the 256 thread sums (Hillis-Steele in LDS)
    uint32_t x = acc;
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t y = (threadIdx.x >= d) ? tsum[threadIdx.x - d] : 0u;
        __syncthreads();
        x += y; tsum[threadIdx.x] = x;
        __syncthreads();
    }
    uint64_t run = block_off[blockIdx.x] + (uint64_t)(x - acc);
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) { if (lo + i < m) out[s0 + lo + i] = run; run += v[i]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = block_off[blockIdx.x] + (uint64_t)x;
}

// ---- write the entries of a batch of groups, key-major --------------------------------------------
// Workgroup = 64 consecutive k-mer codes x 64 groups at a time, transposed through LDS so that a
// wavefront emits one key's entries (<= 64 per step) as one contiguous run.  `cursor` holds, per
// (owner, slot), the position of the key's next entry; it starts as the exclusive scan of the counts
// and is advanced, so batches of groups append in order.
__global__ __launch_bounds__(256) void km_write_kernel(const uint32_t* __restrict__ table, uint64_t T, uint32_t G,
                                                       const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                       uint64_t slots, uint64_t* __restrict__ cursor,
                                                       uint2* __restrict__ entries, uint64_t cap_entries)
{
    __shared__ uint32_t tile[64][65];
    // (cap_entries: `entries` may have been allocated from an estimate, before the host knew the total -- cursor[P * slots], the
    //  end of the scan; a writer that finds less room than that leaves everything untouched and the host runs it again)
    if (cursor[(uint64_t)P * slots] > cap_entries) return;
    const uint64_t x0 = (uint64_t)blockIdx.x * 64;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    // this wavefront's 16 keys x0 + wave + 4 t: their output positions as scalar pointers, advanced by the popcount of each
    // step's ballot (no LDS round trip and no 64-bit lane arithmetic per store)
    size_t cidx = 0;
    uint64_t cur = 0;
    const bool mine = lane < 16 && x0 + wave + 4u * lane < T;
    if (mine) {
        const uint64_t x = x0 + wave + 4u * lane;
        cidx = (size_t)((x % P) * slots + x / P);
        cur = cursor[cidx];
    }
    uint2* dst[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur, t);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur >> 32), t);
        dst[t] = entries + (((uint64_t)hi << 32) | lo);
    }
    const uint32_t xl = threadIdx.x & 63u;
    const bool xok = x0 + xl < T;
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
        // the tile's 16 rows per wavefront: all loads issued before any is used (one at a time, each waited for, this loop was
        // sixteen memory round trips per tile)
        uint32_t v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            const uint32_t gl = wave + 4u * u;
            v[u] = 0;
#ifdef IPK_KMW_NOLOAD             // timing experiment: no table loads (results wrong)
            if (g0 + gl < G && xok) v[u] = (uint32_t)(g0 + gl + x0 + xl) | 0x80000000u;
#else
            if (g0 + gl < G && xok) v[u] = table[(size_t)(g0 + gl) * T + x0 + xl];
#endif
        }
        const uint32_t br = (g0 + lane < G) ? branch_of_group[g0 + lane] : 0u;
        __syncthreads();                                     // the previous tile is consumed
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) tile[wave + 4u * u][xl] = v[u];
        __syncthreads();
        // (the sixteen columns are read before the first store: store8_lanes carries a "memory" clobber, so a read placed between
        //  two stores is issued only after the earlier one -- sixteen exposed LDS round trips per tile instead of one)
        uint32_t wq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) wq[t] = tile[lane][wave + 4u * (uint32_t)t];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t w = wq[t];
            const uint64_t cm = ballot64(w != 0u);
#if defined(IPK_KMW_ALIGNED)      // timing experiment: every store a full, 512-byte aligned one (results wrong)
            store8_lanes(reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(dst[t]) & ~(uintptr_t)511), lane_id() << 3, br, dec_score_bits(w), ~0ull);
#elif defined(IPK_KMW_NOSTORE)    // timing experiment: (almost) no stores (results wrong)
            store8_lanes(dst[t], mbcnt(cm) << 3, br, dec_score_bits(w), cm & (ballot64(w == 0x12345u) != 0 ? ~0ull : 0ull));
#else
            store8_lanes(dst[t], mbcnt(cm) << 3, br, dec_score_bits(w), cm);
#endif
            dst[t] += __popcll(cm);
        }
    }
    // the advanced cursors (batches of groups append in order)
    uint64_t fin = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (lane == (uint32_t)t) fin = (uint64_t)(dst[t] - entries);
    if (mine) cursor[cidx] = fin;
}

// km_write_kernel with every store cut at a 128-byte line of the output.  The dense writer is bound by its stores, and those cost by
// (cache line, instruction) requests and by partly written lines, not by bytes: a key's ~58 entries per 64-group tile are 464 bytes
// at 8-byte alignment -- five lines touched, two of them partly -- and the same stores made full and aligned ran 27 % faster
// (DESIGN.md, section 9).  Here a key's entries that do not fill a line stay behind as a TAIL (< 16 entries, in registers: lane p
// keeps tail entry p of each of the wavefront's 16 keys); per tile the tail and the tile's new entries are compacted into one
// lane-contiguous run through an 80-entry LDS buffer per wavefront (the new entries sit in lanes by group, with holes), the run is
// stored up to its last line boundary -- full, aligned lines only, but for a key's very first and very last store -- and the rest
// becomes the new tail.  Same entries at the same positions as km_write_kernel.
__global__ __launch_bounds__(256) void km_write_lines_kernel(const uint32_t* __restrict__ table, uint64_t T, uint32_t G,
                                                             const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                             uint64_t slots, uint64_t* __restrict__ cursor,
                                                             uint2* __restrict__ entries, uint64_t cap_entries)
{
    __shared__ uint32_t tile[64][65];
    __shared__ uint2 work[4][80];                            // per wavefront: [tail | new entries of the key in hand]
    if (cursor[(uint64_t)P * slots] > cap_entries) return;   // (see km_write_kernel)
    const uint64_t x0 = (uint64_t)blockIdx.x * 64;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    size_t cidx = 0;
    uint64_t cur = 0;
    const bool mine = lane < 16 && x0 + wave + 4u * lane < T;
    if (mine) {
        const uint64_t x = x0 + wave + 4u * lane;
        cidx = (size_t)((x % P) * slots + x / P);
        cur = cursor[cidx];
    }
    uint64_t pos[16];                                        // entry index of the key's next STORE (scalar); its tail follows it
    uint32_t tl[16];                                         // tail lengths (scalar)
    uint32_t tbr[16], tsc[16];                               // lane p: tail entry p of key t
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cur, t);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur >> 32), t);
        pos[t] = ((uint64_t)hi << 32) | lo;
        tl[t] = 0; tbr[t] = 0; tsc[t] = 0;
    }
    // 16 entries = one 128-byte line; `entries` itself is at least 256-byte aligned (hipMalloc / the caching allocator)
    uint2* const wk = work[wave];
    const uint32_t lane8 = lane << 3;
    const uint32_t xl = threadIdx.x & 63u;
    const bool xok = x0 + xl < T;
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
        uint32_t v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            const uint32_t gl = wave + 4u * u;
            v[u] = 0;
            if (g0 + gl < G && xok) v[u] = table[(size_t)(g0 + gl) * T + x0 + xl];
        }
        const uint32_t br = (g0 + lane < G) ? branch_of_group[g0 + lane] : 0u;
        __syncthreads();                                     // the previous tile is consumed
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) tile[wave + 4u * u][xl] = v[u];
        __syncthreads();
        uint32_t wq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) wq[t] = tile[lane][wave + 4u * (uint32_t)t];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t w = wq[t];
            const uint64_t cm = ballot64(w != 0u);
            const uint32_t n = (uint32_t)__popcll(cm);
            const uint32_t tlt = tl[t];
            const uint32_t total = tlt + n;                  // <= 15 + 64
            // run = [tail | new entries], lane-contiguous in LDS
            wave_lds_sync();                                 // the previous key's reads of `wk` are done
            if (lane < tlt) wk[lane] = make_uint2(tbr[t], tsc[t]);
            if (w != 0u) wk[tlt + mbcnt(cm)] = make_uint2(br, dec_score_bits(w));
            wave_lds_sync();
            // stored now: up to the last line boundary at or before the run's end
            const uint32_t over = (uint32_t)((pos[t] + total) & 15u);
            const uint32_t cut = over <= total ? total - over : 0u;          // (no boundary inside the run: all of it stays, total <= 15)
            const uint32_t rem = total - cut;
            const uint2 e0 = wk[lane], e1 = wk[min(64u + lane, 79u)], er = wk[min(cut + lane, 79u)];
            if (cut) {
                uint2* const d = entries + pos[t];
                const uint64_t m0 = cut >= 64 ? ~0ull : ((1ull << cut) - 1ull);
                store8_lanes(d, lane8, e0.x, e0.y, m0);
                if (cut > 64) store8_lanes<512>(d, lane8, e1.x, e1.y, (1ull << (cut - 64)) - 1ull);
            }
            tbr[t] = er.x; tsc[t] = er.y;                    // lanes < rem: the new tail
            tl[t] = rem;
            pos[t] += cut;
        }
    }
    // the tails leave (a key's last, partly filled line) and the cursors advance past them
    uint64_t fin = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (tl[t]) store8_lanes(entries + pos[t], lane8, tbr[t], tsc[t], (1ull << tl[t]) - 1ull);
        if (lane == (uint32_t)t) fin = pos[t] + tl[t];
    }
    if (mine) cursor[cidx] = fin;
}

// km_write_kernel reading the compressed form: the generic version (any number of groups; km_write_c_kernel below is the fast
// one for up to 256 groups per batch).  Workgroup w takes key block (w % 8) * ceil(blocks / 8) + w / 8:
// workgroups are dealt round-robin to the 8 XCDs, so each XCD walks one contiguous range of key blocks and the mask /
// rank lines (eight / sixteen blocks per 64-byte line) are shared in its L2.
__global__ __launch_bounds__(256) void km_write_c_generic_kernel(CompTable ct, uint64_t T, uint32_t G,
                                                         const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                         uint64_t slots, uint64_t* __restrict__ cursor,
                                                         uint2* __restrict__ entries, uint64_t cap_entries)
{
    __shared__ uint32_t tile[64][65];
    __shared__ uint64_t run[64];
    if (cursor[(uint64_t)P * slots] > cap_entries) return;       // (see km_write_kernel)
    __shared__ uint64_t row_mask[64];
    __shared__ const uint32_t* row_vals[64];
    const uint64_t nblocks = (T + 63) / 64, per_xcd = (nblocks + 7) / 8;
    const uint64_t kb = (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || kb >= nblocks) return;
    const uint64_t x0 = kb * 64;
    const uint32_t b = (uint32_t)(x0 / ct.TBL);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    if (threadIdx.x < 64) {
        const uint64_t x = x0 + threadIdx.x;
        run[threadIdx.x] = (x < T) ? cursor[(x % P) * slots + x / P] : 0;
    }
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
        __syncthreads();
        // one thread per group row: the block's occupancy bits and where its values start (three independent loads)
        if (threadIdx.x < 64) {
            const uint32_t g = g0 + threadIdx.x;
            uint64_t m = 0;
            const uint32_t* vp = nullptr;
            if (g < G) {
                const uint32_t* mw = ct.mask + (size_t)g * ct.mask_words + 2 * kb;
                m = (uint64_t)mw[0] | ((uint64_t)mw[1] << 32);
                vp = reinterpret_cast<const uint32_t*>(ct.pool + ct.off[((size_t)g * ct.NB + b) * ct.stride]) +
                     ct.rank[(size_t)g * (ct.mask_words / 2) + kb];
            }
            row_mask[threadIdx.x] = m; row_vals[threadIdx.x] = vp;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 4096; i += 256) {
            const uint32_t gl = i >> 6, xl = i & 63u;
            const uint64_t m = row_mask[gl];
            uint32_t v = 0;
            if ((m >> xl) & 1ull) v = ((global_u32_ptr)row_vals[gl])[(uint32_t)__popcll(m & ((1ull << xl) - 1ull))];
            tile[gl][xl] = v;
        }
        __syncthreads();
        const uint32_t br = (g0 + lane < G) ? branch_of_group[g0 + lane] : 0u;
        for (uint32_t xl = wave; xl < 64; xl += 4) {
            const uint32_t v = tile[lane][xl];
            const uint64_t m = __ballot(v != 0u);
            if (m == 0) continue;
            const uint64_t base = run[xl];
            if (v != 0u) entries[base + mbcnt(m)] = make_uint2(br, dec_score_bits(v));
            if (lane == 0) run[xl] = base + (uint64_t)__popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint64_t x = x0 + threadIdx.x;
        if (x < T) cursor[(x % P) * slots + x / P] = run[threadIdx.x];
    }
}

// A load of read-only data at a wave-uniform address: through the constant address space, so that it becomes a scalar load
// (s_load_*: no vector instruction, the result lands in scalar registers) -- for data a PREVIOUS kernel wrote.
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p)
{
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)(p);      // (via the integer: no generic -> constant cast exists)
}
// the LDS byte address of a __shared__ object (for ds_* instructions written by hand)
__device__ __forceinline__ uint32_t lds_address(const void* p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(p);
}

// dec_score_bits without a select (the select form inside the copy loops below crashes this compiler's instruction selection)
__device__ __forceinline__ uint32_t dec_score_bits_bf(uint32_t e) { return e ^ ~((uint32_t)((int32_t)e >> 31) >> 1); }

#ifndef IPK_KMC_CAP
#define IPK_KMC_CAP 4224
#endif
#ifndef IPK_KMC_SB
#define IPK_KMC_SB 8
#endif
#ifndef IPK_KMC_VB
#define IPK_KMC_VB 8
#endif
#ifndef IPK_KMC_RL
#define IPK_KMC_RL 1             // a row's bits and value address by v_readlane (1) or by a 16-byte broadcast LDS read (0) in the value-load phase
#endif
constexpr uint32_t KMC_CAP = IPK_KMC_CAP;   // entries of a key block staged in LDS at once (33 KiB + 6 KiB of row data: four workgroups per CU.
                                            // While the kernel needed 150+ VGPRs -- three wavefronts per SIMD -- a smaller stage bought nothing
                                            // (3840 / 2816: equal at a cfg3 share, 13-17 % slower at cfg4); at 113 VGPRs 4224 is 2-3 % ahead of 5632
                                            // at both, 3328 equal to 4224)

// km_write_kernel reading the compressed form, for batches of up to 256 groups.  Workgroup w takes key block
// (w % 8) * ceil(blocks / 8) + w / 8: workgroups are dealt round-robin to the 8 XCDs, so each XCD walks one contiguous range
// of key blocks and the lines of the occupancy bits / value addresses (eight blocks per 64-byte line) are shared in its L2.
//
// The work is a transposition: the values of a (group, block) are contiguous in memory in KEY order, a key's entries are
// contiguous in the output in GROUP order.  The round-2 profile of the tile version (km_write_c_generic_kernel: a 64 x 64
// tile in LDS per 64 groups, two barriers per tile) showed it bound by instruction issue and by the latency of three small
// scattered loads per row, at 1.2x the algorithmic HBM bytes but a quarter of the bandwidth.  This version spends ~5 vector
// instructions per ROW and keeps a wavefront's 64 value loads in flight together:
//   rows     one (group, block) row at a time with the 64 KEYS across the lanes: the row's occupancy bits and the address of
//            its values are wave-uniform (scalar loads), lane x's value is values[popcount(bits below x)] -- one coalesced load;
//   scatter  the entry goes to LDS at (key's offset in the block's output) + (groups before this one that hold the key):
//            a per-lane running position, advanced under exec = the row's bits (written to exec directly: no compares);
//   split    the four wavefronts take a quarter of the rows each (<= 64); where each starts, per key, comes from the
//            per-quarter counts the counting kernel leaves in qpack;
//   copy     the staged block leaves as one linear, fully coalesced copy (scores decoded here, where all lanes are busy).
// A block whose entries exceed the LDS stage is done in several key ranges (greedy split at key boundaries).
// (Measured at cfg4, 19.3 ms: 14.2 without the value loads, 14.7 without the stores (IPK_KMC_NOLOAD / IPK_KMC_NOSTORE builds) -- neither
//  memory side is the bound alone; rows broadcast from lane registers with v_readlane instead of the LDS round trip: 25.9 ms, and
//  4.46 against 3.56 ms at a cfg3 share -- seven v_readlane per row cost more than the three broadcast reads they replace; bits and
//  branch id of a row in ONE 16-byte broadcast read for the scatter (three LDS instructions per row instead of four, stage 5248):
//  3.54 ms at the cfg3 share, 20.4 at cfg4 -- the LDS pipe is not the bound either.)
template <bool ONE_OWNER, uint32_t CAP>
__global__ __launch_bounds__(256) void km_write_c_kernel(CompTable ct, uint64_t T, uint32_t G,
                                                         const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                         uint64_t slots, const uint32_t* __restrict__ counts,
                                                         const uint32_t* __restrict__ qpack,
                                                         uint64_t* __restrict__ cursor, uint2* __restrict__ entries, uint64_t cap_entries)
{
    __shared__ uint2 out[CAP];
    if (cursor[(uint64_t)P * slots] > cap_entries) return;       // (see km_write_kernel)
#if !IPK_KMC_RL
    __shared__ uint4 rowmeta[4][64];                         // per wavefront and row: occupancy bits, address of the values
#endif
    __shared__ uint32_t rowbr[4][64];                        //                        branch id
    __shared__ uint32_t kpre[65];                            // exclusive prefix of the block's per-key entry counts
    __shared__ uint64_t kcur[64];                            // the keys' output positions
    const uint64_t nblocks = (T + 63) / 64, per_xcd = (nblocks + 7) / 8;
    const uint64_t kb = (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || kb >= nblocks) return;
    const uint64_t x0 = kb * 64;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();

    // this wavefront's rows [r0, r0 + nrows), nrows <= 64.  Their occupancy bits, value addresses and branch ids come in with
    // ONE coalesced-as-can-be load each (lane = row) and are staged in LDS, from where every row's are read back as broadcasts:
    // the per-row values are wave-uniform without a chain of scalar-load round trips (a first version that fetched them with
    // s_load, eight rows at a time, spent its time waiting for those).  The loads are issued first thing, together with the keys'
    // counts and cursors below: one round trip for all of them.
    const uint32_t rq = (G + 3) / 4, r0 = min(G, wave * rq), nrows = min(G, r0 + rq) - r0;
    uint64_t row_m = 0, row_va = reinterpret_cast<uint64_t>(ct.pool);                    // (rows past the end: no bits, any readable address)
    uint32_t row_br = 0;
    {
        const uint32_t row_bytes = (uint32_t)(ct.mask_words / 2) * 8u;                   // a row of bits / of addresses
        const bool live = lane < nrows;
        const size_t ro = (size_t)(r0 + (live ? lane : 0u)) * row_bytes;
        if (live) {
            row_m = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(reinterpret_cast<const uint64_t*>(ct.mask) + kb) + ro);
            row_va = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(ct.vaddr + kb) + ro);
            row_br = branch_of_group[r0 + lane];
        }
    }

    // this lane's key: where it lives, and how many rows of the quarters before this wavefront's hold it
    const uint64_t x = x0 + lane;
    const size_t cidx = x < T ? (ONE_OWNER ? (size_t)x : (size_t)((x % P) * slots + x / P)) : 0;
    const uint32_t qp = x < T ? qpack[cidx] : 0u;
    uint32_t before = 0;
#pragma unroll
    for (uint32_t w = 0; w < 3; ++w) before += w < wave ? (qp >> (8u * w)) & 0xFFu : 0u;
    uint32_t my_cnt = 0;
    if (threadIdx.x < 64) {
        uint64_t cur = 0;
        if (x < T) { my_cnt = counts[cidx]; cur = cursor[cidx]; }
        kcur[threadIdx.x] = cur;
        uint32_t inc = my_cnt;
        for (uint32_t o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
        kpre[threadIdx.x + 1] = inc;
        if (threadIdx.x == 0) kpre[0] = 0;
    }

#if !IPK_KMC_RL
    rowmeta[wave][lane] = make_uint4((uint32_t)row_m, (uint32_t)(row_m >> 32), (uint32_t)row_va, (uint32_t)(row_va >> 32));
#endif
    rowbr[wave][lane] = row_br;
    // the value loads of all rows, none waited for here.  Every lane loads: a lane without the key reads the value of the next
    // key that has one (or, past the row's end, whatever follows it in the pool, which is allocated 256 B longer for this) and
    // does not store it.
    uint32_t val[64];
    const uint32_t rm_lo = (uint32_t)row_m, rm_hi = (uint32_t)(row_m >> 32), ra_lo = (uint32_t)row_va, ra_hi = (uint32_t)(row_va >> 32);
#if IPK_KMC_RL
    // (round 4: a row's bits and value address reach the lanes by v_readlane from the lane-per-row registers, not by a 16-byte
    //  broadcast read -- r04_cfg4_backhalf_sq.json has the LDS pipe busy 85 % of this kernel's time, and of a row's ~18 LDS cycles
    //  eight were that read: 64 lanes x 16 bytes come back through the LDS data path whether the address is shared or not)
#pragma unroll
    for (uint32_t b8 = 0; b8 < 64; b8 += 8) {
        if (b8 < nrows) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t r = b8 + u;
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)rm_lo, (int)r), mhi = (uint32_t)__builtin_amdgcn_readlane((int)rm_hi, (int)r);
                const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)ra_lo, (int)r), ahi = (uint32_t)__builtin_amdgcn_readlane((int)ra_hi, (int)r);
                const uint32_t j = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                const global_u32_ptr vals = (global_u32_ptr)(((uint64_t)ahi << 32) | alo);
                val[r] = __builtin_nontemporal_load(vals + j);
            }
        }
    }
#else
#pragma unroll
    for (uint32_t b8 = 0; b8 < 64; b8 += 8) {
        if (b8 < nrows) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                if constexpr (IPK_KMC_VB < 8) { if ((u % IPK_KMC_VB) == 0) asm volatile("" ::: "memory"); }   // (bounds the rows whose reads are in flight: registers)
                const uint4 mv = rowmeta[wave][b8 + u];                                  // (same address in every lane: a broadcast read)
                const uint32_t j = __builtin_amdgcn_mbcnt_hi(mv.y, __builtin_amdgcn_mbcnt_lo(mv.x, 0u));
                // (a GLOBAL load: through a generic pointer this was flat_load_dword, which counts on lgkmcnt as well -- the wait for
                //  the next eight rows' broadcast reads then waited for these eight loads too, eight memory round trips per wavefront
                //  one after the other)
                const global_u32_ptr vals = (global_u32_ptr)(((uint64_t)mv.w << 32) | mv.z);
#ifdef IPK_KMC_NOLOAD            // timing experiment: no value loads (results wrong)
                val[b8 + u] = j + mv.z;
#else
                val[b8 + u] = __builtin_nontemporal_load(vals + j);
#endif
            }
        }
    }
#endif
    __syncthreads();
    const uint32_t out_lds = lds_address(out);

    for (uint32_t ka = 0; ka < 64;) {
        // the key range [ka, ke) of this round: as many keys as fit the stage
        const uint32_t pre_a = kpre[ka];
        const uint32_t fits = (uint32_t)__popcll(ballot64(lane >= ka && kpre[lane + 1] - pre_a <= CAP));
        const uint32_t ke = ka + max(fits, 1u);                                         // (G <= CAP: one key always fits)
        const uint32_t n_part = kpre[ke] - pre_a;
        if (n_part == 0) { ka = ke; continue; }
        const uint64_t pm = (ke >= 64 ? ~0ull : ((1ull << ke) - 1ull)) & ~((1ull << ka) - 1ull);
        const uint32_t pm_lo = (uint32_t)pm, pm_hi = (uint32_t)(pm >> 32);

        // the rows' entries into the stage  (lanes outside the range never store; a whole block needs no masking)
        uint32_t posb = out_lds + (kpre[lane] - pre_a + before) * 8u;
        auto scatter = [&](auto WHOLE, uint32_t& pb) {                                  // (pb: an asm operand must not be a capture)
            constexpr uint32_t SB = IPK_KMC_SB;                                          // rows whose bits and branch ids are read together
#pragma unroll
            for (uint32_t b8 = 0; b8 < 64; b8 += 8) {
                if (b8 < nrows) {
#pragma unroll
                    for (uint32_t b4 = 0; b4 < 8; b4 += SB) {
                        // (the rows' bits and branch ids are read before the first of the asm statements: those carry a "memory"
                        //  clobber, so a read placed between them is issued only after the previous row's write -- one exposed LDS round
                        //  trip per row instead of one per SB rows)
                        uint2 mmq[SB]; uint32_t brq[SB];
                        asm volatile("" ::: "memory");                   // (fresh reads: reusing the value-load phase's copies would keep 128 registers alive)
#pragma unroll
                        for (uint32_t u = 0; u < SB; ++u) {
#if IPK_KMC_RL
                            mmq[u] = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rm_lo, (int)(b8 + b4 + u)),
                                                (uint32_t)__builtin_amdgcn_readlane((int)rm_hi, (int)(b8 + b4 + u)));     // (no LDS read: see the value-load phase)
#else
                            mmq[u] = *reinterpret_cast<const uint2*>(&rowmeta[wave][b8 + b4 + u]);
#endif
#if IPK_KMC_RL >= 2
                            brq[u] = (uint32_t)__builtin_amdgcn_readlane((int)row_br, (int)(b8 + b4 + u));
#else
                            brq[u] = rowbr[wave][b8 + b4 + u];
#endif
                        }
#pragma unroll
                        for (uint32_t u = 0; u < SB; ++u) {
                            uint2 mm = mmq[u];
                            if constexpr (!decltype(WHOLE)::value) { mm.x &= pm_lo; mm.y &= pm_hi; }
                            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)mm.x);
                            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)mm.y);
                            IPK_ASSERT_FULL_EXEC();
                            // (branch and score as two 32-bit operands of one ds_write2_b32: as a 64-bit pair the compiler parks every
                            //  val[] in the high half of a register pair -- 176 VGPRs, two wavefronts per SIMD)
                            const uint32_t ex = brq[u], ey = val[b8 + b4 + u];          // (locals: an asm operand must not be a capture)
                            asm volatile("s_mov_b64 exec, %3\n\tds_write2_b32 %0, %1, %2 offset1:1\n\tv_add_u32 %0, 8, %0\n\ts_mov_b64 exec, -1"
                                         : "+v"(pb) : "v"(ex), "v"(ey), "s"(((uint64_t)hi << 32) | lo) : "memory");
                        }
                    }
                }
            }
        };
        if (pm == ~0ull) scatter(std::true_type{}, posb); else scatter(std::false_type{}, posb);
        // the scatter's ds_write_b64 live inside asm statements, which the compiler's waitcnt pass does not count: without this
        // wait the barrier below is a bare s_barrier, and s_barrier does not wait for outstanding LDS operations -- the other
        // wavefronts could read `out` before this one's writes have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // copy out
        bool linear = ONE_OWNER;
        if (ONE_OWNER) {                                     // (a fresh scan makes consecutive keys adjacent; checked, not assumed)
            const bool ok = lane < ka || lane + 1 >= ke || kcur[lane] + (kpre[lane + 1] - kpre[lane]) == kcur[lane + 1];
            linear = ballot64(ok) == ~0ull;
        }
        if (linear) {
            uint2* dst = entries + kcur[ka];
#pragma unroll 2
            for (uint32_t i = threadIdx.x; i < n_part; i += 256) {
                uint2 e = out[i];
                e.y = dec_score_bits_bf(e.y);
#ifdef IPK_KMC_NOSTORE           // timing experiment: (almost) no stores (results wrong)
                if (e.x == 0xFFFFFFF1u) dst[i] = e;
#else
                dst[i] = e;
#endif
            }
        } else {
            for (uint32_t t = ka + wave; t < ke; t += 4) {
                const uint32_t n = kpre[t + 1] - kpre[t];
                const uint2* src = out + (kpre[t] - pre_a);
                uint2* dst = entries + kcur[t];
                for (uint32_t i = lane; i < n; i += 64) {
                    uint2 e = src[i];
                    e.y = dec_score_bits_bf(e.y);
                    dst[i] = e;
                }
            }
        }
        ka = ke;
        if (ka < 64) __syncthreads();                        // the stage is reused
    }
    // the advanced cursors (batches of groups append in order)
    if (threadIdx.x < 64 && x < T) cursor[cidx] = kcur[threadIdx.x] + my_cnt;
}

// km_write_c_kernel walking a RUN of up to KMC_RUN consecutive key blocks of one bucket slice (round 4).
//
// After the rows' bits and addresses stopped going through LDS, the vector-memory path was what the writer waited for
// (r04b_cfg4_backhalf_sq.json: the L1 "pending" 85 % of the time, address unit stalled by the cache 38 %, 909 64-byte read requests
// per key block): more than half of those requests were the lane-per-row loads of a block's bits and value addresses -- 8 useful
// bytes out of every line, 512 lines per workgroup.  Consecutive blocks of a row are consecutive in memory, and inside a bucket
// slice the next block's values follow this block's (vaddr[g][blk + 1] = vaddr[g][blk] + 4 * popcount(bits[g][blk]),
// compress_slice), so a workgroup that walks R consecutive blocks loads a row's bits for all of them at once (R x 8 contiguous
// bytes per lane) and a row's value address ONCE: 2 / R requests per row and block instead of 2.  The blocks are done one after the
// other with the stage, the key prefix and everything else of km_write_c_kernel; the next block's per-key counts and cursors are
// requested while this block is copied out.
#ifndef IPK_KMC_RUN
#define IPK_KMC_RUN 4
#endif
#ifndef IPK_KMC_RUN_NT
#define IPK_KMC_RUN_NT 1         // value loads of the run form with the non-temporal hint (1, as km_write_c_kernel) or without (0).  Where the run form is used
                                 // (up to 128 groups) the hint is equal or slightly ahead (cfg3 share, three alternations: 2.81-2.87 against 2.82-3.10 ms);
                                 // forced at cfg4 it loses -- a line shared by two blocks of a run is read again microseconds later by the SAME workgroup, and
                                 // evict-first lines were gone by then (L2 hit rate 43 % against 64 %, HBM reads + 58 %: 15.6 against 14.6 ms)
#endif
#ifndef IPK_KMC_WPE
#define IPK_KMC_WPE 1            // 1: registers bounded for four wavefronts per SIMD (four workgroups per CU, as km_write_c_kernel)
#endif
#if IPK_KMC_WPE
#define KMC_RUN_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define KMC_RUN_OCCUPANCY
#endif
constexpr uint32_t KMC_RUN = IPK_KMC_RUN;
__host__ __device__ inline uint64_t kmc_runs(uint64_t T, uint32_t TBL)
{
    const uint64_t nblocks = (T + 63) / 64, bpb = TBL / 64, nbuckets = (nblocks + bpb - 1) / bpb;
    return nbuckets * ((bpb + KMC_RUN - 1) / KMC_RUN);
}
template <bool ONE_OWNER, uint32_t CAP>
__global__ __launch_bounds__(256) KMC_RUN_OCCUPANCY void km_write_c_run_kernel(CompTable ct, uint64_t T, uint32_t G,
                                                             const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                             uint64_t slots, const uint32_t* __restrict__ counts,
                                                             const uint32_t* __restrict__ qpack,
                                                             uint64_t* __restrict__ cursor, uint2* __restrict__ entries, uint64_t cap_entries)
{
    __shared__ uint2 out[CAP];
    if (cursor[(uint64_t)P * slots] > cap_entries) return;       // (see km_write_kernel)
    __shared__ uint32_t rowbr[4][64];                        // per wavefront and row: branch id
    __shared__ uint32_t kpre[65];                            // exclusive prefix of the block's per-key entry counts
    __shared__ uint64_t kcur[64];                            // the keys' output positions
    const uint64_t nblocks = (T + 63) / 64, bpb = ct.TBL / 64, rpb = (bpb + KMC_RUN - 1) / KMC_RUN;
    const uint64_t nruns = kmc_runs(T, ct.TBL), per_xcd = (nruns + 7) / 8;
    const uint64_t run = (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);    // (each XCD walks a contiguous range of runs)
    if ((blockIdx.x >> 3) >= per_xcd || run >= nruns) return;
    const uint64_t bucket = run / rpb;
    const uint64_t kb0 = bucket * bpb + (run - bucket * rpb) * KMC_RUN;
    if (kb0 >= nblocks) return;
    const uint32_t nb = (uint32_t)(min(min(kb0 + KMC_RUN, (bucket + 1) * bpb), nblocks) - kb0);
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();

    // this wavefront's rows [r0, r0 + nrows), nrows <= 64: lane = row.  The bits of all the run's blocks, the value address of the
    // first block, the branch id.
    const uint32_t rq = (G + 3) / 4, r0 = min(G, wave * rq), nrows = min(G, r0 + rq) - r0;
    uint64_t mq[KMC_RUN];
    uint64_t row_va = reinterpret_cast<uint64_t>(ct.pool);                                // (rows past the end: no bits, any readable address)
    uint32_t row_br = 0;
    {
        const uint32_t row_bytes = (uint32_t)(ct.mask_words / 2) * 8u;                   // a row of bits / of addresses
        const bool live = lane < nrows;
        const size_t ro = (size_t)(r0 + (live ? lane : 0u)) * row_bytes;
        const uint64_t* mrow = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(reinterpret_cast<const uint64_t*>(ct.mask) + kb0) + ro);
#pragma unroll
        for (uint32_t i = 0; i < KMC_RUN; ++i) mq[i] = (live && i < nb) ? mrow[i] : 0ull;
        if (live) {
            row_va = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(ct.vaddr + kb0) + ro);
            row_br = branch_of_group[r0 + lane];
        }
    }
    rowbr[wave][lane] = row_br;

    // the first block's keys: where they live, how many rows of the quarters before this wavefront's hold them
    auto key_index = [&](uint64_t x) -> size_t { return x < T ? (ONE_OWNER ? (size_t)x : (size_t)((x % P) * slots + x / P)) : 0; };
    size_t cidx = key_index(kb0 * 64 + lane);
    uint32_t qp = kb0 * 64 + lane < T ? qpack[cidx] : 0u;
    uint32_t my_cnt = 0;
    uint64_t cur = 0;
    if (threadIdx.x < 64 && kb0 * 64 + lane < T) { my_cnt = counts[cidx]; cur = cursor[cidx]; }
    const uint32_t out_lds = lds_address(out);

    for (uint32_t it = 0; it < nb; ++it) {
        const uint64_t x = (kb0 + it) * 64 + lane;
        const uint64_t row_m = mq[0];
        uint32_t before = 0;
#pragma unroll
        for (uint32_t w = 0; w < 3; ++w) before += w < wave ? (qp >> (8u * w)) & 0xFFu : 0u;
        if (threadIdx.x < 64) {
            kcur[threadIdx.x] = cur;
            uint32_t inc = my_cnt;
            for (uint32_t o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
            kpre[threadIdx.x + 1] = inc;
            if (threadIdx.x == 0) kpre[0] = 0;
        }
        const uint32_t cnt_now = my_cnt;

        // the value loads of all rows, none waited for here.  Every lane loads: a lane without the key reads the value of the next
        // key that has one (or, past the row's end, whatever follows it in the pool, which is allocated 256 B longer for this) and
        // does not store it.
        uint32_t val[64];
        const uint32_t rm_lo = (uint32_t)row_m, rm_hi = (uint32_t)(row_m >> 32), ra_lo = (uint32_t)row_va, ra_hi = (uint32_t)(row_va >> 32);
#pragma unroll
        for (uint32_t b8 = 0; b8 < 64; b8 += 8) {
            if (b8 < nrows) {
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t r = b8 + u;
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)rm_lo, (int)r), mhi = (uint32_t)__builtin_amdgcn_readlane((int)rm_hi, (int)r);
                    const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)ra_lo, (int)r), ahi = (uint32_t)__builtin_amdgcn_readlane((int)ra_hi, (int)r);
                    const uint32_t j = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                    const global_u32_ptr vals = (global_u32_ptr)(((uint64_t)ahi << 32) | alo);
                    val[r] = IPK_KMC_RUN_NT ? __builtin_nontemporal_load(vals + j) : vals[j];
                }
            }
        }
        // the next block of the run: its values follow this block's; its bits move up; its keys' counts and cursors are requested now
        row_va += 4ull * (uint32_t)__popcll(row_m);
#pragma unroll
        for (uint32_t i = 0; i + 1 < KMC_RUN; ++i) mq[i] = mq[i + 1];
        mq[KMC_RUN - 1] = 0;
        if (it + 1 < nb) {
            const uint64_t xn = x + 64;
            cidx = key_index(xn);
            qp = xn < T ? qpack[cidx] : 0u;
            my_cnt = 0; cur = 0;
            if (threadIdx.x < 64 && xn < T) { my_cnt = counts[cidx]; cur = cursor[cidx]; }
        }
        __syncthreads();

        for (uint32_t ka = 0; ka < 64;) {
            // the key range [ka, ke) of this round: as many keys as fit the stage
            const uint32_t pre_a = kpre[ka];
            const uint32_t fits = (uint32_t)__popcll(ballot64(lane >= ka && kpre[lane + 1] - pre_a <= CAP));
            const uint32_t ke = ka + max(fits, 1u);                                         // (G <= CAP: one key always fits)
            const uint32_t n_part = kpre[ke] - pre_a;
            if (n_part == 0) { ka = ke; continue; }
            const uint64_t pm = (ke >= 64 ? ~0ull : ((1ull << ke) - 1ull)) & ~((1ull << ka) - 1ull);
            const uint32_t pm_lo = (uint32_t)pm, pm_hi = (uint32_t)(pm >> 32);

            // the rows' entries into the stage  (lanes outside the range never store; a whole block needs no masking)
            uint32_t posb = out_lds + (kpre[lane] - pre_a + before) * 8u;
            auto scatter = [&](auto WHOLE, uint32_t& pb) {                                  // (pb: an asm operand must not be a capture)
                constexpr uint32_t SB = IPK_KMC_SB;                                          // rows whose branch ids are read together
#pragma unroll
                for (uint32_t b8 = 0; b8 < 64; b8 += 8) {
                    if (b8 < nrows) {
#pragma unroll
                        for (uint32_t b4 = 0; b4 < 8; b4 += SB) {
                            uint2 mmq[SB]; uint32_t brq[SB];
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (uint32_t u = 0; u < SB; ++u) {
                                mmq[u] = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)rm_lo, (int)(b8 + b4 + u)),
                                                    (uint32_t)__builtin_amdgcn_readlane((int)rm_hi, (int)(b8 + b4 + u)));
                                brq[u] = rowbr[wave][b8 + b4 + u];
                            }
#pragma unroll
                            for (uint32_t u = 0; u < SB; ++u) {
                                uint2 mm = mmq[u];
                                if constexpr (!decltype(WHOLE)::value) { mm.x &= pm_lo; mm.y &= pm_hi; }
                                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)mm.x);
                                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)mm.y);
                                IPK_ASSERT_FULL_EXEC();
                                const uint32_t ex = brq[u], ey = val[b8 + b4 + u];          // (locals: an asm operand must not be a capture)
                                asm volatile("s_mov_b64 exec, %3\n\tds_write2_b32 %0, %1, %2 offset1:1\n\tv_add_u32 %0, 8, %0\n\ts_mov_b64 exec, -1"
                                             : "+v"(pb) : "v"(ex), "v"(ey), "s"(((uint64_t)hi << 32) | lo) : "memory");
                            }
                        }
                    }
                }
            };
            if (pm == ~0ull) scatter(std::true_type{}, posb); else scatter(std::false_type{}, posb);
            // (the scatter's writes live inside asm statements the compiler's wait-count pass does not see: see km_write_c_kernel)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();

            // copy out
            bool linear = ONE_OWNER;
            if (ONE_OWNER) {                                     // (a fresh scan makes consecutive keys adjacent; checked, not assumed)
                const bool ok = lane < ka || lane + 1 >= ke || kcur[lane] + (kpre[lane + 1] - kpre[lane]) == kcur[lane + 1];
                linear = ballot64(ok) == ~0ull;
            }
            if (linear) {
                uint2* dst = entries + kcur[ka];
#pragma unroll 2
                for (uint32_t i = threadIdx.x; i < n_part; i += 256) {
                    uint2 e = out[i];
                    e.y = dec_score_bits_bf(e.y);
                    dst[i] = e;
                }
            } else {
                for (uint32_t t = ka + wave; t < ke; t += 4) {
                    const uint32_t n = kpre[t + 1] - kpre[t];
                    const uint2* src = out + (kpre[t] - pre_a);
                    uint2* dst = entries + kcur[t];
                    for (uint32_t i = lane; i < n; i += 64) {
                        uint2 e = src[i];
                        e.y = dec_score_bits_bf(e.y);
                        dst[i] = e;
                    }
                }
            }
            ka = ke;
            if (ka < 64) __syncthreads();                        // the stage is reused
        }
        // the advanced cursors (batches of groups append in order)
        if (threadIdx.x < 64 && x < T) cursor[key_index(x)] = kcur[threadIdx.x] + cnt_now;
        __syncthreads();                                         // stage, key prefix and cursors are the next block's now
    }
}

// ---- merge of S sources for one owner -----------------------------------------------------------
// Every source brings its own counts row [slots] and its own entry block: counts[s], src[s] are device pointers.
__global__ __launch_bounds__(256) void merge_sum_counts_kernel(const uint32_t* const* __restrict__ counts, uint32_t S,
                                                               uint64_t slots, uint32_t* __restrict__ total,
                                                               uint32_t* __restrict__ flags)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= slots) return;
    uint32_t c = 0;
    for (uint32_t s = 0; s < S; ++s) c += ((global_u32_ptr)counts[s])[q];
    total[q] = c;
    flags[q] = (c != 0u);
}

// One wavefront per slot: copies the slot's entries of every source, in source order.
// goff = ONE exclusive scan over the sources' counts rows laid end to end ([S * slots + 1]): source s's entries of slot q start at
// goff[s * slots + q] - goff[s * slots] inside the source's own block.  Lane s fetches source s's count, offset and base in one
// round trip (sixty-four sources at a time); the copies then run source by source with everything they need in registers --
// the first version walked the sources with three dependent loads each (pointer, count, offset) before the copy could start.
__global__ __launch_bounds__(256) void merge_copy_kernel(const uint32_t* const* __restrict__ counts, uint32_t S, uint64_t slots,
                                                         const uint64_t* __restrict__ goff,      // [S * slots + 1]
                                                         const uint2* const* __restrict__ src,
                                                         const uint64_t* __restrict__ dst_off,   // [slots+1]
                                                         uint2* __restrict__ dst,
                                                         uint64_t first)                         // (a launch covers slots from `first`: grid x block stays below 2^32)
{
    const uint64_t q = first + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= slots) return;
    const uint32_t lane = lane_id();
    uint64_t d = dst_off[q];
    for (uint32_t s0 = 0; s0 < S; s0 += 64) {
        const uint32_t s = s0 + lane;
        uint32_t n_s = 0;
        uint64_t from_s = 0;
        if (s < S) {
            const global_u32_ptr row = (global_u32_ptr)counts[s];
            n_s = row[q];
            from_s = reinterpret_cast<uint64_t>(src[s]) + 8ull * (goff[(uint64_t)s * slots + q] - goff[(uint64_t)s * slots]);
        }
        const uint32_t ns = min(64u, S - s0);
        // four sources per trip: their loads are issued together, then their stores
        for (uint32_t t0 = 0; t0 < ns; t0 += 4) {
            typedef const unsigned long long __attribute__((address_space(1)))* global_u64_ptr;   // (an entry as one 64-bit word)
            unsigned long long* const dst64 = reinterpret_cast<unsigned long long*>(dst);
            uint32_t n[4]; global_u64_ptr from[4]; unsigned long long v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t t = min(t0 + u, ns - 1);
                n[u] = t0 + u < ns ? (uint32_t)__builtin_amdgcn_readlane((int)n_s, (int)t) : 0u;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)from_s, (int)t);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(from_s >> 32), (int)t);
                from[u] = (global_u64_ptr)(((uint64_t)hi << 32) | lo);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) if (lane < n[u]) v[u] = from[u][lane];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (lane < n[u]) dst64[d + lane] = v[u];
                for (uint32_t i = lane + 64; i < n[u]; i += 64) dst64[d + i] = from[u][i];   // (more than 64 entries of one source: rare)
                d += n[u];
            }
        }
    }
}

// single source: total = counts, flags = (counts != 0)
__global__ __launch_bounds__(256) void flags_from_counts_kernel(const uint32_t* __restrict__ counts, uint64_t slots,
                                                                uint32_t* __restrict__ total, uint32_t* __restrict__ flags)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= slots) return;
    const uint32_t c = counts[q];
    total[q] = c;
    flags[q] = (c != 0u);
}

__global__ void add_base_kernel(uint64_t* __restrict__ v, uint64_t n, uint64_t base)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += base;
}

template <int SIGMA>
__device__ __forceinline__ uint32_t pack_code_km(uint32_t dense, int k)
{
    if constexpr (SIGMA == 4) { (void)k; return dense; }
    else {
        uint32_t key = 0;
        for (int d = 0; d < k; ++d) {
            const uint32_t q = dense / SIGMA;
            key |= (dense - q * SIGMA) << (5 * d);
            dense = q;
        }
        return key;
    }
}

// keys[j], key_off[j] for the j-th non-empty slot (flag_off = exclusive scan of flags)
template <int SIGMA>
__global__ __launch_bounds__(256) void merge_write_keys_kernel(const uint32_t* __restrict__ total,
                                                               const uint64_t* __restrict__ flag_off,
                                                               const uint64_t* __restrict__ dst_off, uint64_t slots,
                                                               uint32_t owner, uint32_t P, int k,
                                                               uint32_t* __restrict__ keys, uint64_t* __restrict__ key_off)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q > slots) return;
    if (q == slots) { key_off[flag_off[slots]] = dst_off[slots]; return; }
    if (total[q] != 0u) {
        const uint64_t j = flag_off[q];
        keys[j] = pack_code_km<SIGMA>((uint32_t)(q * P + owner), k);
        key_off[j] = dst_off[q];
    }
}

}  // namespace ipkgpu
