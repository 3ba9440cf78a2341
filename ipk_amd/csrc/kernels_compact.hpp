// kernels_compact.hpp -- dense per-group tables -> group-major CSR (sorted keys per branch group).
#pragma once
#include "dcla_device.hpp"
#include "comp_table.hpp"

namespace ipkgpu {

// ---- table -> CSR compaction -------------------------------------------------------------------
constexpr uint32_t CHUNK = 4096;   // table slots per workgroup

__global__ __launch_bounds__(256) void count_chunks_kernel(const uint32_t* __restrict__ table, uint64_t table_size,
                                                           uint32_t chunks_per_group, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.x / chunks_per_group, c = blockIdx.x - g * chunks_per_group;
    const uint32_t* t = table + (size_t)g * table_size;
    const uint64_t s0 = (uint64_t)c * CHUNK;
    const uint32_t n = (uint32_t)min((uint64_t)CHUNK, table_size - s0);
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) cnt += (t[s0 + i] != 0u);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Exclusive scan of n u32 counts into n+1 u64 offsets (single workgroup; n is ~1e5..1e7).
__global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t* __restrict__ counts, uint64_t n,
                                                           uint64_t base, uint64_t* __restrict__ offsets)
{
    __shared__ uint64_t part[1024];
    const uint64_t per = (n + 1023) / 1024;
    const uint64_t lo = min(n, (uint64_t)threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = base;
        for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = acc; acc += v; }
        offsets[n] = acc;
    }
    __syncthreads();
    uint64_t acc = part[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { offsets[i] = acc; acc += counts[i]; }
}

// offsets[g * stride] for g in [0, n) -> out[g]: the per-group CSR offsets of a batch
__global__ void gather_offsets_kernel(const uint64_t* __restrict__ offsets, uint32_t stride, uint32_t n,
                                      uint64_t* __restrict__ out)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) out[g] = offsets[(size_t)g * stride];
}

template <int SIGMA>
__device__ __forceinline__ uint32_t pack_code(uint32_t dense, int k)
{
    if constexpr (SIGMA == 4) { (void)k; return dense; }
    else {
        constexpr int BITS = 5;
        uint32_t key = 0;
        for (int d = 0; d < k; ++d) {                 // last symbol in the lowest bits
            const uint32_t q = dense / SIGMA;
            key |= (dense - q * SIGMA) << (BITS * d);
            dense = q;
        }
        return key;
    }
}

template <int SIGMA>
__global__ __launch_bounds__(256) void write_chunks_kernel(const uint32_t* __restrict__ table, uint64_t table_size,
                                                           uint32_t chunks_per_group, int k,
                                                           const uint64_t* __restrict__ offsets,
                                                           uint32_t* __restrict__ keys, float* __restrict__ scores)
{
    __shared__ uint32_t wcnt[4];
    const uint32_t g = blockIdx.x / chunks_per_group, c = blockIdx.x - g * chunks_per_group;
    const uint32_t* t = table + (size_t)g * table_size;
    const uint64_t s0 = (uint64_t)c * CHUNK;
    const uint32_t n = (uint32_t)min((uint64_t)CHUNK, table_size - s0);
    uint64_t out = offsets[blockIdx.x];
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t v = 0;
        if (i < n) v = t[s0 + i];
        const uint64_t m = __ballot(v != 0u);
        if (lane_id() == 0) wcnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) { const uint32_t x = wcnt[q]; total += x; if (q < wave) before += x; }
        if (v != 0u) {
            const uint64_t pos = out + before + mbcnt(m);
            keys[pos] = pack_code<SIGMA>((uint32_t)(s0 + i), k);
            scores[pos] = __uint_as_float(dec_score_bits(v));
        }
        out += total;
        __syncthreads();
    }
}


// ---- the same two steps from the occupancy bits / the compressed table form ---------------------------
// counts[g * chunks_per_group + c] from the mask (128 words per chunk of 4096 slots)
__global__ __launch_bounds__(256) void count_chunks_mask_kernel(const uint32_t* __restrict__ mask, uint64_t mask_words,
                                                                uint32_t chunks_per_group, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.x / chunks_per_group, c = blockIdx.x - g * chunks_per_group;
    const uint64_t w0 = (uint64_t)c * (CHUNK / 32);
    uint32_t cnt = 0;
    if (threadIdx.x < CHUNK / 32 && w0 + threadIdx.x < mask_words) cnt = (uint32_t)__popc(mask[(size_t)g * mask_words + w0 + threadIdx.x]);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Group-major CSR straight from the compressed form: a (group, bucket) slice already holds its non-empty slots' scores in
// key order, and the slices of a group follow each other in key order, so the CSR position of a slice is the exclusive
// scan of the slices' entry counts (ucnt).  One workgroup per slice: keys from the occupancy bits, scores decoded.
template <int SIGMA>
__global__ __launch_bounds__(256) void write_group_c_kernel(CompTable ct, uint64_t table_size, int k,
                                                            const uint64_t* __restrict__ offsets,      // [slices + 1]
                                                            uint32_t* __restrict__ keys, float* __restrict__ scores)
{
    constexpr uint32_t MAXBLK = 512;                                   // TBL <= 32768 slots
    __shared__ uint64_t smask[MAXBLK];
    __shared__ uint32_t srank[MAXBLK];
    const uint32_t gb = blockIdx.x, g = gb / ct.NB, b = gb - g * ct.NB;
    const uint64_t key0 = (uint64_t)b * ct.TBL;
    const uint32_t nslots = (uint32_t)min((uint64_t)ct.TBL, table_size - key0), nblk = (nslots + 63) / 64;
    const uint32_t* __restrict__ vals = reinterpret_cast<const uint32_t*>(ct.pool + ct.off[(size_t)gb * ct.stride]);
    const uint32_t* mrow = ct.mask + (size_t)g * ct.mask_words + (key0 >> 5);
    const uint32_t* rrow = ct.rank + (size_t)g * (ct.mask_words / 2) + (key0 >> 6);
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) {               // the slice's bits and ranks: coalesced, once
        smask[i] = (uint64_t)mrow[2 * i] | ((uint64_t)mrow[2 * i + 1] << 32);
        srank[i] = rrow[i];
    }
    __syncthreads();
    const uint64_t out = offsets[gb];
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    // eight blocks per trip: their value loads are in flight together
    constexpr uint32_t UB = 8;
    for (uint32_t blk0 = wave * UB; blk0 < nblk; blk0 += 4 * UB) {
        uint32_t pos[UB], v[UB];
        bool has[UB];
#pragma unroll
        for (uint32_t u = 0; u < UB; ++u) {
            const uint32_t blk = blk0 + u;
            const uint64_t m = blk < nblk ? smask[blk] : 0ull;
            has[u] = (m >> lane) & 1ull;
            pos[u] = (blk < nblk ? srank[blk] : 0u) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            v[u] = has[u] ? vals[pos[u]] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < UB; ++u)
            if (has[u]) {
                keys[out + pos[u]] = pack_code<SIGMA>((uint32_t)(key0 + (blk0 + u) * 64 + lane), k);
                scores[out + pos[u]] = __uint_as_float(dec_score_bits(v[u]));
            }
    }
}

// ---- positions variant: 64-bit table (score code << 32 | ~sequence) -> keys, scores, positions -----
__global__ __launch_bounds__(256) void count_chunks64_kernel(const unsigned long long* __restrict__ table, uint64_t table_size,
                                                             uint32_t chunks_per_group, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.x / chunks_per_group, c = blockIdx.x - g * chunks_per_group;
    const unsigned long long* t = table + (size_t)g * table_size;
    const uint64_t s0 = (uint64_t)c * CHUNK;
    const uint32_t n = (uint32_t)min((uint64_t)CHUNK, table_size - s0);
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) cnt += (t[s0 + i] != 0ull);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <int SIGMA>
__global__ __launch_bounds__(256) void write_chunks_pos_kernel(const unsigned long long* __restrict__ table, uint64_t table_size,
                                                               uint32_t chunks_per_group, int k, uint32_t nwin,
                                                               const uint64_t* __restrict__ offsets, uint32_t* __restrict__ keys,
                                                               float* __restrict__ scores, uint32_t* __restrict__ positions)
{
    __shared__ uint32_t wcnt[4];
    const uint32_t g = blockIdx.x / chunks_per_group, c = blockIdx.x - g * chunks_per_group;
    const unsigned long long* t = table + (size_t)g * table_size;
    const uint64_t s0 = (uint64_t)c * CHUNK;
    const uint32_t n = (uint32_t)min((uint64_t)CHUNK, table_size - s0);
    uint64_t out = offsets[blockIdx.x];
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        unsigned long long v = 0;
        if (i < n) v = t[s0 + i];
        const uint64_t m = __ballot(v != 0ull);
        if (lane_id() == 0) wcnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) { const uint32_t x = wcnt[q]; total += x; if (q < wave) before += x; }
        if (v != 0ull) {
            const uint64_t pos = out + before + mbcnt(m);
            keys[pos] = pack_code<SIGMA>((uint32_t)(s0 + i), k);
            scores[pos] = __uint_as_float(dec_score_bits((uint32_t)(v >> 32)));
            positions[pos] = (0xFFFFFFFFu - (uint32_t)v) % nwin;      // window start inside its matrix (window::get_position)
        }
        out += total;
        __syncthreads();
    }
}

}  // namespace ipkgpu
