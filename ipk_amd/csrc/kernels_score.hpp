// kernels_score.hpp -- prefix array + scoring kernels (see ipkgpu.hip for the data layout).
#pragma once
#include "dcla_device.hpp"

namespace ipkgpu {


// ---- matrix::preprocess (window.cpp:16-27): best[0] = 0, best[j+1] = best[j] + max_i m[j][i] ---
// One workgroup per matrix.  Column maxima are computed by all lanes (coalesced), the running sum
// is accumulated by ONE lane in site order -- a parallel scan would round differently, and the
// rounding noise of this array is part of the reference semantics.
template <int SIGMA>
__global__ __launch_bounds__(256) void prefix_max_kernel(const float* __restrict__ logp, uint32_t sites,
                                                         float* __restrict__ best)
{
    constexpr int CH = 4096;
    __shared__ float cm[CH];
    __shared__ float carry;
    const uint32_t mat = blockIdx.x;
    const float* m = logp + (size_t)mat * sites * SIGMA;
    float* b = best + (size_t)mat * (sites + 1);
    if (threadIdx.x == 0) { carry = 0.0f; b[0] = 0.0f; }
    for (uint32_t c0 = 0; c0 < sites; c0 += CH) {
        const uint32_t n = min((uint32_t)CH, sites - c0);
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            const float4* col = reinterpret_cast<const float4*>(m + (size_t)(c0 + j) * SIGMA);
            float largest;
            {
                const float4 v = col[0];
                largest = v.x;                                  // std::max_element: first largest
                if (largest < v.y) largest = v.y;
                if (largest < v.z) largest = v.z;
                if (largest < v.w) largest = v.w;
            }
#pragma unroll
            for (int q = 1; q < SIGMA / 4; ++q) {
                const float4 v = col[q];
                if (largest < v.x) largest = v.x;
                if (largest < v.y) largest = v.y;
                if (largest < v.z) largest = v.z;
                if (largest < v.w) largest = v.w;
            }
            cm[j] = largest;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float acc = carry;
#pragma unroll 8
            for (uint32_t j = 0; j < n; ++j) { acc += cm[j]; cm[j] = acc; }
            carry = acc;
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) b[c0 + j + 1] = cm[j];
        __syncthreads();
    }
}

struct ScoreParams {
    const float* logp;            // [n_mats][sites][SIGMA]
    const float* best;            // [n_mats][sites+1]
    const uint32_t* mat_list;     // matrices of this batch (indices into logp/best)
    const uint32_t* mat_slot;     // [n_mats] table slot of each matrix within this batch
    uint32_t n_batch_mats, sites, nwin, tiles_per_mat;
    float eps;
    uint32_t* table;              // [slots][table_size]
    uint64_t table_size;
    unsigned long long* emitted;
    unsigned long long* ovf_queue; // (mat << 32 | window start) of windows whose lists overflowed
    uint32_t* ovf_count;
    uint32_t flags;               // bit 0 (diagnostic builds of bench only): skip the table update
};

// One window: build both half lists, then the final filtered cross product goes straight into the
// group's max table (ipk::put, branch_group.cpp:88-101).  Returns false if a list overflowed CAP
// (nothing has been emitted for the window in that case).
template <int SIGMA, int K, int CAP>
__device__ __forceinline__ bool score_window(const WinCtx& c, float eps, uint2* scratch,
                                             uint32_t* __restrict__ tab, uint32_t& emitted, bool no_put = false)
{
    if constexpr (Geo<SIGMA, K, CAP>::DIRECT) {
        const uint32_t lane = lane_id();
        float s = 0.f;
        bool pass = false;
        if (lane < Geo<SIGMA, K, CAP>::FULL) pass = Direct<SIGMA, 0, K>::eval(c, eps, lane, s);
        if (pass) atomicMax(tab + lane, enc_score_bits(__float_as_uint(s)));
        emitted += (uint32_t)__popcll(__ballot(pass));
        return true;
    } else {
        const uint2 *L, *R;
        uint32_t nL, nR;
        if (!build_halves<SIGMA, K, CAP>(c, eps, scratch, L, nL, R, nR)) return false;
        if (nL == 0 || nR == 0) return true;
        constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
        uint32_t cnt = 0;
        for_each_pair(L, nL, R, nR, [&](bool valid, uint2 a, uint2 b) {
            const float s = __uint_as_float(a.y) + __uint_as_float(b.y);      // pk_compute.cpp:90
            const bool pass = valid && (s > eps);                              // :91
            if (pass && !no_put) atomicMax(tab + (a.x * mulR + b.x), enc_score_bits(__float_as_uint(s)));
            cnt += (uint32_t)__popcll(__ballot(pass));
        });
        emitted += cnt;
        return true;
    }
}

template <int SIGMA, int K, int TW>
struct TileGeo {
    static constexpr int TC = TW + K - 1;                 // columns a tile of TW windows touches
    static constexpr int COLS_F = TC * SIGMA;             // floats (multiple of 4)
    static constexpr int BEST_F = ((TC + 1 + 3) / 4) * 4;
    static constexpr int HEAD_BYTES = (COLS_F + BEST_F) * 4;
};

// Fast path: a workgroup stages the columns of TW consecutive windows of one matrix in LDS
// (coalesced 16-byte loads), its NW wavefronts take windows round-robin.
template <int SIGMA, int K, int CAP, int TW, int NW>
__global__ __launch_bounds__(NW * 64) void score_tiles_kernel(ScoreParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    using TG = TileGeo<SIGMA, K, TW>;
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch_all = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    constexpr uint32_t WS = wave_scratch_entries<SIGMA, K, CAP>();

    const uint32_t bm = blockIdx.x / p.tiles_per_mat;
    const uint32_t tile = blockIdx.x - bm * p.tiles_per_mat;
    const uint32_t mat = p.mat_list[bm];
    const uint32_t t0 = tile * TW;
    const uint32_t nw = min((uint32_t)TW, p.nwin - t0);
    const uint32_t ncol = nw + K - 1;

    {
        const float4* src = reinterpret_cast<const float4*>(p.logp + ((size_t)mat * p.sites + t0) * SIGMA);
        float4* dst = reinterpret_cast<float4*>(cols);
        const uint32_t n4 = ncol * (SIGMA / 4);
        for (uint32_t i = threadIdx.x; i < n4; i += NW * 64) dst[i] = src[i];
        const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + t0;
        for (uint32_t i = threadIdx.x; i <= ncol; i += NW * 64) best[i] = bsrc[i];
    }
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6;
    uint2* scratch = scratch_all + (size_t)wave * WS;
    uint32_t* tab = p.table + (size_t)p.mat_slot[mat] * p.table_size;
    uint32_t emitted = 0;
    for (uint32_t w = wave; w < nw; w += NW) {
        WinCtx c{cols, best, w};
        if (!score_window<SIGMA, K, CAP>(c, p.eps, scratch, tab, emitted, (p.flags & 1u) != 0)) {
            if (lane_id() == 0) {
                const uint32_t q = atomicAdd(p.ovf_count, 1u);
                p.ovf_queue[q] = ((unsigned long long)mat << 32) | (unsigned long long)(t0 + w);
            }
        }
    }
    if (lane_id() == 0 && emitted) atomicAdd(p.emitted, (unsigned long long)emitted);
}

// Big-list path: one wavefront per workgroup with worst-case list capacity (sigma^(k/2) entries per
// half list), walking the queue of windows the fast path could not hold.  Every wave reaches the
// loop exit: the queue length is fixed before this kernel starts.
template <int SIGMA, int K>
__global__ __launch_bounds__(64) void score_overflow_kernel(ScoreParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int CAPF = 1 << 30;
    using TG = TileGeo<SIGMA, K, 1>;
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    const uint32_t n = *p.ovf_count;
    uint32_t emitted = 0;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const unsigned long long e = p.ovf_queue[q];
        const uint32_t mat = (uint32_t)(e >> 32), start = (uint32_t)e;
        const float* src = p.logp + ((size_t)mat * p.sites + start) * SIGMA;
        for (uint32_t i = threadIdx.x; i < K * SIGMA; i += 64) cols[i] = src[i];
        const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + start;
        for (uint32_t i = threadIdx.x; i <= K; i += 64) best[i] = bsrc[i];
        wave_lds_sync();
        WinCtx c{cols, best, 0};
        uint32_t* tab = p.table + (size_t)p.mat_slot[mat] * p.table_size;
        score_window<SIGMA, K, CAPF>(c, p.eps, scratch, tab, emitted);
        wave_lds_sync();
    }
    if (lane_id() == 0 && emitted) atomicAdd(p.emitted, (unsigned long long)emitted);
}


}  // namespace ipkgpu
