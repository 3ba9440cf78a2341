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
#include "dcla_device.hpp"
#include "comp_table.hpp"

namespace ipkgpu {

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
                                                            uint32_t G, uint32_t P, uint64_t slots, uint32_t* __restrict__ counts)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w * 32 >= T) return;
    uint32_t plane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t total[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) total[b] = 0;
    const uint32_t* m = mask + w;
    for (uint32_t g0 = 0; g0 < G; g0 += 255) {
        const uint32_t g1 = min(G, g0 + 255u);
        for (uint32_t g = g0; g < g1; ++g) {
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
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) plane[p] = 0;
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) {
        const uint64_t x = w * 32 + b;
        if (x < T) counts[(x % P) * slots + x / P] = total[b];
    }
}

// One thread per key: for key spaces too small to fill the chip with one thread per mask word (DNA k <= 11).
__global__ __launch_bounds__(256) void km_count_mask_key_kernel(const uint32_t* __restrict__ mask, uint64_t W, uint64_t T,
                                                                uint32_t G, uint32_t P, uint64_t slots, uint32_t* __restrict__ counts)
{
    const uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= T) return;
    const uint32_t* m = mask + (x >> 5);
    const uint32_t sh = (uint32_t)x & 31u;
    uint32_t c = 0, g = 0;
    for (; g + 4 <= G; g += 4) {
        const uint32_t v0 = m[(size_t)(g + 0) * W], v1 = m[(size_t)(g + 1) * W];
        const uint32_t v2 = m[(size_t)(g + 2) * W], v3 = m[(size_t)(g + 3) * W];
        c += ((v0 >> sh) & 1u) + ((v1 >> sh) & 1u) + ((v2 >> sh) & 1u) + ((v3 >> sh) & 1u);
    }
    for (; g < G; ++g) c += (m[(size_t)g * W] >> sh) & 1u;
    const uint64_t o = x % P, q = x / P;
    counts[o * slots + q] = c;
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
                                                       uint2* __restrict__ entries)
{
    __shared__ uint32_t tile[64][65];
    __shared__ uint64_t run[64];
    const uint64_t x0 = (uint64_t)blockIdx.x * 64;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    if (threadIdx.x < 64) {
        const uint64_t x = x0 + threadIdx.x;
        run[threadIdx.x] = (x < T) ? cursor[(x % P) * slots + x / P] : 0;
    }
    for (uint32_t g0 = 0; g0 < G; g0 += 64) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 4096; i += 256) {
            const uint32_t gl = i >> 6, xl = i & 63u;
            uint32_t v = 0;
            if (g0 + gl < G && x0 + xl < T) v = table[(size_t)(g0 + gl) * T + x0 + xl];
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

// km_write_kernel reading the compressed form.  Workgroup w takes key block (w % 8) * ceil(blocks / 8) + w / 8:
// workgroups are dealt round-robin to the 8 XCDs, so each XCD walks one contiguous range of key blocks and the mask /
// rank lines (eight / sixteen blocks per 64-byte line) are shared in its L2.
__global__ __launch_bounds__(256) void km_write_c_kernel(CompTable ct, uint64_t T, uint32_t G,
                                                         const uint32_t* __restrict__ branch_of_group, uint32_t P,
                                                         uint64_t slots, uint64_t* __restrict__ cursor,
                                                         uint2* __restrict__ entries)
{
    __shared__ uint32_t tile[64][65];
    __shared__ uint64_t run[64];
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
            if ((m >> xl) & 1ull) v = row_vals[gl][(uint32_t)__popcll(m & ((1ull << xl) - 1ull))];
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

// ---- merge of S sources for one owner -----------------------------------------------------------
// Every source brings its own counts row [slots] and its own entry block: counts[s], src[s] are device pointers.
__global__ __launch_bounds__(256) void merge_sum_counts_kernel(const uint32_t* const* __restrict__ counts, uint32_t S,
                                                               uint64_t slots, uint32_t* __restrict__ total,
                                                               uint32_t* __restrict__ flags)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= slots) return;
    uint32_t c = 0;
    for (uint32_t s = 0; s < S; ++s) c += counts[s][q];
    total[q] = c;
    flags[q] = (c != 0u);
}

// One wavefront per slot: copies the slot's entries of every source, in source order.
// src_off[s][q] = exclusive scan of source s's counts (relative to the source's own block).
__global__ __launch_bounds__(256) void merge_copy_kernel(const uint32_t* const* __restrict__ counts, uint32_t S, uint64_t slots,
                                                         const uint64_t* __restrict__ src_off,   // [S][slots+1]
                                                         const uint2* const* __restrict__ src,
                                                         const uint64_t* __restrict__ dst_off,   // [slots+1]
                                                         uint2* __restrict__ dst)
{
    const uint64_t q = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= slots) return;
    const uint32_t lane = lane_id();
    uint64_t d = dst_off[q];
    for (uint32_t s = 0; s < S; ++s) {
        const uint32_t n = counts[s][q];
        const uint2* from = src[s] + src_off[(size_t)s * (slots + 1) + q];
        for (uint32_t i = lane; i < n; i += 64) dst[d + i] = from[i];
        d += n;
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
