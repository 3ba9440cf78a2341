// kernels_score.hpp -- prefix array + scoring kernels (see ipkgpu.hip for the data layout).
#pragma once
#include <type_traits>
#include "dcla_device.hpp"

namespace ipkgpu {


// ---- matrix::preprocess (window.cpp:16-27): best[0] = 0, best[j+1] = best[j] + max_i m[j][i] ---
// Column maxima are computed by all lanes (coalesced), the running sum is accumulated by ONE lane per matrix in site
// order -- a parallel scan would round differently, and the rounding noise of this array is part of the reference
// semantics.  A workgroup takes M consecutive matrices: their chains run side by side in lanes
// 0..M-1 of wavefront 0 (a chain is latency, 4 ns per addition whatever the other lanes do), so that a call's matrices
// fit one round of workgroups -- with a workgroup per matrix cfg2's 2000 were 7.8 per CU against 7 resident, two rounds:
// 0.22 ms where a rank's share of 250 took 0.065.
template <int SIGMA, int M>
__global__ __launch_bounds__(256) void prefix_max_kernel(const float* __restrict__ logp, uint32_t n_mats, uint32_t sites,
                                                         float* __restrict__ best)
{
    // Two buffers of column maxima: while the chain lanes add their way through one chunk of sites, wavefronts 1..3 compute the
    // next chunk's maxima (round 4: the chain used to wait for every chunk's maxima, all 256 threads for the chain in turn).
    constexpr int CH = 2048 / M;
    constexpr int ROW = CH + 4;                              // (the chain lanes' 16-byte reads of their own rows: banks 4 lanes apart)
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "matrices per workgroup");
    __shared__ __align__(16) float cmbuf[2][M][ROW];
    const uint32_t mat0 = blockIdx.x * M;
    const uint32_t mv = min((uint32_t)M, n_mats - mat0);     // matrices of this workgroup (host: blockIdx.x * M < n_mats)
    const bool chain = threadIdx.x < mv;
    const uint32_t my = mat0 + (chain ? threadIdx.x : 0u);   // the chain lane's matrix
    float* b = best + (size_t)my * (sites + 1);
    if (chain) b[0] = 0.0f;
    float carry = 0.0f;                                      // (the chain lanes')
    auto maxima = [&](uint32_t c0, float (*cmw)[ROW], uint32_t t0, uint32_t nt) {
        const uint32_t n = min((uint32_t)CH, sites - c0);
        for (uint32_t mm = 0; mm < mv; ++mm) {
            const float* m = logp + ((size_t)(mat0 + mm) * sites + c0) * SIGMA;
            for (uint32_t j = t0; j < n; j += nt) {
                const float4* col = reinterpret_cast<const float4*>(m + (size_t)j * SIGMA);
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
                cmw[mm][j] = largest;
            }
        }
    };
    if (sites > 0) maxima(0, cmbuf[0], threadIdx.x, blockDim.x);
    uint32_t ci = 0;
    for (uint32_t c0 = 0; c0 < sites; c0 += CH, ++ci) {
        const uint32_t n = min((uint32_t)CH, sites - c0);
        __syncthreads();                                   // this chunk's maxima are in place; the other buffer is free
        if (threadIdx.x >= 64) {
            if (c0 + CH < sites) maxima(c0 + CH, cmbuf[(ci + 1) & 1], threadIdx.x - 64, blockDim.x - 64);
        } else
        if (chain) {
            // The chain of float additions IS the algorithm (a scan would round differently).  A bare chain of dependent v_add_f32 costs
            // 4.0 ns per addition on this part (tools/micro_addchain.hip); until round 4 this loop took 9.5-10: the sums went back to LDS
            // and every trip waited for those writes (s_waitcnt lgkmcnt(0)) before its next reads.  Now the adding lane only READS LDS --
            // sixteen maxima per round trip, three register sets so that a set's reads are two sets (128 ns of additions) old when they
            // are needed -- and stores the sums straight to best[] (plain 4-byte stores: nothing ever waits for them).
            const float* cm = cmbuf[ci & 1][threadIdx.x];
            float acc = carry;
            const float4* c4 = reinterpret_cast<const float4*>(cm);
            float* dst = b + c0 + 1;
            const uint32_t n16 = n / 16;
            auto add16 = [&](const float4& v0, const float4& v1, const float4& v2, const float4& v3, float* o) {
                float s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, sa, sb, sc, sd, se, sf;
                s0 = acc + v0.x; s1 = s0 + v0.y; s2 = s1 + v0.z; s3 = s2 + v0.w;
                s4 = s3 + v1.x; s5 = s4 + v1.y; s6 = s5 + v1.z; s7 = s6 + v1.w;
                s8 = s7 + v2.x; s9 = s8 + v2.y; sa = s9 + v2.z; sb = sa + v2.w;
                sc = sb + v3.x; sd = sc + v3.y; se = sd + v3.z; sf = se + v3.w;
                acc = sf;
                o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = s5; o[6] = s6; o[7] = s7;
                o[8] = s8; o[9] = s9; o[10] = sa; o[11] = sb; o[12] = sc; o[13] = sd; o[14] = se; o[15] = sf;
            };
            float4 r[3][4];
#pragma unroll
            for (uint32_t s_ = 0; s_ < 2; ++s_)
                if (s_ < n16) { r[s_][0] = c4[4 * s_]; r[s_][1] = c4[4 * s_ + 1]; r[s_][2] = c4[4 * s_ + 2]; r[s_][3] = c4[4 * s_ + 3]; }
            uint32_t q = 0;
            for (; q + 3 <= n16; q += 3) {
#pragma unroll
                for (uint32_t u = 0; u < 3; ++u) {
                    const uint32_t nx = q + u + 2;                               // the set two ahead goes into the registers just freed
                    if (nx < n16) { r[(u + 2) % 3][0] = c4[4 * nx]; r[(u + 2) % 3][1] = c4[4 * nx + 1]; r[(u + 2) % 3][2] = c4[4 * nx + 2]; r[(u + 2) % 3][3] = c4[4 * nx + 3]; }
                    add16(r[u][0], r[u][1], r[u][2], r[u][3], dst + 16 * (q + u));
                }
            }
            // (q is a multiple of 3: the one or two sets left sit in r[0], r[1], their reads issued above)
            if (q < n16) add16(r[0][0], r[0][1], r[0][2], r[0][3], dst + 16 * q);
            if (q + 1 < n16) add16(r[1][0], r[1][1], r[1][2], r[1][3], dst + 16 * (q + 1));
            for (uint32_t j = n16 * 16; j < n; ++j) { acc += cm[j]; dst[j] = acc; }
            carry = acc;
        }
    }
}

struct ScoreParams {
    const float* logp;            // [n_mats][sites][SIGMA]
    const float* best;            // [n_mats][sites+1]
    const uint32_t* mat_list;     // matrices of this batch (indices into logp/best)
    const uint32_t* mat_slot;     // [n_mats] table slot of each matrix within this batch
    uint32_t n_batch_mats, sites, nwin, tiles_per_mat;
    float eps;
    void* table;                  // [slots][table_size] u32 score codes; u64 (code << 32 | ~sequence) when positions are kept
    uint64_t table_size;
    const uint32_t* mat_rank;     // [n_mats] rank of each matrix inside its group (positions variant only)
    unsigned long long* emitted;
    unsigned long long* ovf_queue; // (mat << 32 | window start) of windows whose lists overflowed
    uint32_t* ovf_count;
    uint32_t flags;               // bit 0 (diagnostic builds of bench only): skip the table update
    uint32_t* mask;               // [slots][mask_words] occupancy bits of the tables (bit x % 32 of word x / 32), or null:
    uint64_t mask_words;          //   written by the LDS reduce passes, kept current by the big-list kernel
    uint32_t* big_ovf = nullptr;  // set when a window's half list exceeds the big-list kernels' capped capacity (big_capf, DNA k >= 13)
};

// ipk::put (branch_group.cpp:88-101): keep the larger score; the first one wins ties.
struct PutScore {
    uint32_t* tab;
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t score_bits) const
    {
        atomicMax(tab + idx, enc_score_bits(score_bits));
    }
};
// KEEP_POSITIONS variant (branch_group.cpp:73-86, db_builder.cpp:655-662): the window position travels with
// the score.  64-bit max over (score code << 32 | ~sequence), sequence = processing order of the window inside
// its group (matrix rank * windows + start): larger score wins, equal scores keep the EARLIER window.
struct PutScorePos {
    unsigned long long* tab;
    uint32_t inv_seq;
    __device__ __forceinline__ void operator()(uint32_t idx, uint32_t score_bits) const
    {
        atomicMax(tab + idx, ((unsigned long long)enc_score_bits(score_bits) << 32) | (unsigned long long)inv_seq);
    }
};

// One window: build both half lists, then the final filtered cross product goes straight into the
// group's max table.  Returns false if a list overflowed CAP (nothing has been emitted for the window
// in that case).
template <int SIGMA, int K, int CAP, class Put>
__device__ __forceinline__ bool score_window(const WinCtx& c, float eps, uint2* scratch,
                                             const Put& put, unsigned long long& emitted, bool no_put = false)
{
    if constexpr (Geo<SIGMA, K, CAP>::DIRECT) {
        const uint32_t lane = lane_id();
        float s = 0.f;
        bool pass = false;
        if (lane < Geo<SIGMA, K, CAP>::FULL) pass = Direct<SIGMA, 0, K>::eval(c, eps, lane, s);
        if (pass && !no_put) put(lane, __float_as_uint(s));
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
            if (pass && !no_put) put(a.x * mulR + b.x, __float_as_uint(s));
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
template <int SIGMA, int K, int CAP, int TW, int NW, bool POS = false>
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
    const size_t tab_off = (size_t)p.mat_slot[mat] * p.table_size;
    unsigned long long emitted = 0;                     // per-wave count of scored phylo-k-mers (can pass 2^32 on flat data)
    for (uint32_t w = wave; w < nw; w += NW) {
        WinCtx c{cols, best, w};
        bool ok;
        if constexpr (POS) {
            const PutScorePos put{reinterpret_cast<unsigned long long*>(p.table) + tab_off,
                                  0xFFFFFFFFu - (p.mat_rank[mat] * p.nwin + t0 + w)};
            ok = score_window<SIGMA, K, CAP>(c, p.eps, scratch, put, emitted, (p.flags & 1u) != 0);
        } else {
            const PutScore put{reinterpret_cast<uint32_t*>(p.table) + tab_off};
            ok = score_window<SIGMA, K, CAP>(c, p.eps, scratch, put, emitted, (p.flags & 1u) != 0);
        }
        if (!ok) {
            if (lane_id() == 0) {
                const uint32_t q = atomicAdd(p.ovf_count, 1u);
                p.ovf_queue[q] = ((unsigned long long)mat << 32) | (unsigned long long)(t0 + w);
            }
        }
    }
    if (lane_id() == 0 && emitted) atomicAdd(p.emitted, emitted);
}

// Big-list path: windows whose half lists exceed the fast path's capacity.  One workgroup of OVF_NW
// wavefronts per window with worst-case list capacity (sigma^ceil(k/2) entries per half list): wave 0
// builds the two half lists, then all waves share the final cross product (it is what is big here:
// |L| x |R| up to 4096^2) and max-reduce into the group's table with global atomics.  Every wave
// reaches the loop exit: the queue length is fixed before this kernel starts.
#ifndef IPK_OVF_NW
#define IPK_OVF_NW 8
#endif
constexpr int OVF_NW = IPK_OVF_NW;
#ifndef IPK_OVF_UNCOND_OR
#define IPK_OVF_UNCOND_OR 0        // 1: no-return atomicMax + an unconditional atomicOr on the occupancy bits (measured: see DESIGN A.4)
#endif
#ifndef IPK_OVF_PEEK
#define IPK_OVF_PEEK 0
#endif
#ifndef IPK_OVF_NOPUT
#define IPK_OVF_NOPUT 0            // 1: the big-list kernel without its table atomics (timing experiments; results wrong)
#endif
// Capacity of the big-list kernels' half lists.  Up to DNA k = 12 / AA k = 6 a window's worst-case lists (sigma^ceil(k/2) entries
// each) fit LDS and nothing can overflow; from DNA k = 13 they do not (2 x 4^7 x 8 B = 256 KB), so the lists are capped at
// BIG_CAP_ENTRIES and a window whose half list exceeds that raises *big_ovf: the call fails loudly (IPKGPU_ERR_INVALID) instead of
// dropping k-mers.  (Flat posteriors at low omega can get there; the reference has no such limit -- its lists live in host memory.)
constexpr int BIG_CAP_ENTRIES = 6144;
template <int SIGMA, int K> constexpr int big_capf()
{
    return (size_t)wave_scratch_entries<SIGMA, K, 1 << 30>() * 8 <= (size_t)136 * 1024 ? (1 << 30) : BIG_CAP_ENTRIES;
}

template <int SIGMA, int K, bool POS = false>
__global__ __launch_bounds__(OVF_NW * 64) void score_overflow_kernel(ScoreParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int CAPF = big_capf<SIGMA, K>();
    using TG = TileGeo<SIGMA, K, 1>;
    __shared__ uint32_t sh_n[2];
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    const uint32_t n = *p.ovf_count;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long emitted = 0;                     // per-wave count of scored phylo-k-mers (can pass 2^32 on flat data)
    constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const unsigned long long e = p.ovf_queue[q];
        const uint32_t mat = (uint32_t)(e >> 32), start = (uint32_t)e;
        __syncthreads();                                                       // previous window's lists consumed
        const float* src = p.logp + ((size_t)mat * p.sites + start) * SIGMA;
        for (uint32_t i = threadIdx.x; i < K * SIGMA; i += OVF_NW * 64) cols[i] = src[i];
        const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + start;
        for (uint32_t i = threadIdx.x; i <= K; i += OVF_NW * 64) best[i] = bsrc[i];
        __syncthreads();
        const size_t tab_off = (size_t)p.mat_slot[mat] * p.table_size;
        uint32_t* tab = reinterpret_cast<uint32_t*>(p.table) + tab_off;
        unsigned long long* tab64 = reinterpret_cast<unsigned long long*>(p.table) + tab_off;
        const uint32_t inv_seq = POS ? 0xFFFFFFFFu - (p.mat_rank[mat] * p.nwin + start) : 0u;
        const uint2 *L = scratch, *R = scratch + Geo<SIGMA, K / 2, CAPF>::CAPH;
        if (wave == 0) {
            WinCtx c{cols, best, 0};
            uint32_t nL = 0, nR = 0;
            const bool fits = build_halves<SIGMA, K, CAPF>(c, p.eps, scratch, L, nL, R, nR);      // (cannot overflow at full capacity)
            if (!fits) { nL = 0; nR = 0; if (lane == 0 && p.big_ovf) atomicOr(p.big_ovf, 1u); }   // capped lists (big_capf): the call fails
            if (lane == 0) { sh_n[0] = nL; sh_n[1] = nR; }
        }
        __syncthreads();
        const uint32_t nL = sh_n[0], nR = sh_n[1];
        if (nL == 0 || nR == 0) continue;
        uint32_t cnt = 0;
        // A block of 64 entries of R stays in registers while the rows of L (dealt round-robin to the waves) pass by, OVF_ROWS rows per
        // trip with their LDS reads in flight together.  (Until round 4 every (row, block) step read both operands from LDS and waited
        // for them: ~400 cycles per step, and the step count -- |L| x |R| / 64, 1 % of them pairs -- is what this kernel's time is:
        // without its atomics it took as long.)
        constexpr uint32_t OVF_ROWS = 4;
        const bool no_put = (p.flags & 1u) != 0 || IPK_OVF_NOPUT;
        for (uint32_t jb = 0; jb < nR; jb += 64) {
            const uint32_t j = jb + lane;
            const bool valid = j < nR;
            uint2 b = make_uint2(0, 0);
            if (valid) b = R[j];
            const float by = __uint_as_float(b.y);
            for (uint32_t i0 = wave; i0 < nL; i0 += OVF_NW * OVF_ROWS) {
                uint2 a[OVF_ROWS];
#pragma unroll
                for (uint32_t u = 0; u < OVF_ROWS; ++u) a[u] = L[min(i0 + u * OVF_NW, nL - 1)];
#pragma unroll
                for (uint32_t u = 0; u < OVF_ROWS; ++u) {
                    const float s = __uint_as_float(a[u].y) + by;                     // pk_compute.cpp:90
                    const bool pass = valid && (i0 + u * OVF_NW < nL) && (s > p.eps);  // :91
                    if (pass && !no_put) {
                        const uint32_t idx = a[u].x * mulR + b.x;
                        if constexpr (POS) PutScorePos{tab64, inv_seq}(idx, __float_as_uint(s));
                        else {
                            // the returned old value says whether this is the slot's first score (measured cheaper than a plain
                            // read of the mask word followed by a conditional atomicOr: 0.62 vs 0.74 ms at cfg2)
#if IPK_OVF_UNCOND_OR
                            atomicMax(tab + idx, enc_score_bits(__float_as_uint(s)));                        // (no-return forms: nothing waits)
                            if (p.mask) atomicOr(p.mask + (size_t)p.mat_slot[mat] * p.mask_words + (idx >> 5), 1u << (idx & 31u));
#elif IPK_OVF_PEEK
                            // a plain look first: the slot's value only ever grows, so a score that does not beat what is there needs no atomic
                            const uint32_t enc = enc_score_bits(__float_as_uint(s));
                            if (__builtin_nontemporal_load(tab + idx) < enc) {
                                const uint32_t old = atomicMax(tab + idx, enc);                              // PutScore
                                if (old == 0u && p.mask)
                                    atomicOr(p.mask + (size_t)p.mat_slot[mat] * p.mask_words + (idx >> 5), 1u << (idx & 31u));
                            }
#else
                            const uint32_t old = atomicMax(tab + idx, enc_score_bits(__float_as_uint(s)));   // PutScore
                            if (old == 0u && p.mask)
                                atomicOr(p.mask + (size_t)p.mat_slot[mat] * p.mask_words + (idx >> 5), 1u << (idx & 31u));
#endif
                        }
                    }
                    cnt += (uint32_t)__popcll(__ballot(pass));
                }
            }
        }
        emitted += cnt;
    }
    if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);
}

// =================================================================================================
// Two-pass radix max-reduce ("stream" variant): scattered global atomics run at ~25-45 G/s on MI355X
// (memory-side, 64-byte requests) while LDS atomics run at ~1.8 T/s, so the per-branch max-reduce
// (ipk::put, branch_group.cpp:88-101) is done in LDS.  A group's table (sigma^k slots) does not fit
// LDS, hence:
//   pass 1  score_stream_kernel: list building as in score_tiles_kernel; every surviving
//           (code, score) pair is appended to a chunk of the pair pool owned by this wavefront and
//           the pair's key bucket (bucket = code / TBL).  Chunks are CH pairs; a descriptor
//           (group, bucket, count) is written when a chunk is closed.
//   index   chunk_hist / scan / chunk_scatter: chunk ids sorted by (group, bucket).
//   pass 2  reduce_buckets_kernel: one workgroup per (group, bucket) max-reduces its chunks in a
//           TBL-slot LDS table and stores the finished table slice (plain coalesced stores).
// =================================================================================================
// Pairs per chunk of the pair pool.  A bucket's open chunk is rolled every CH pairs, and a roll is ~50 instructions plus a trip
// through the slow path of the append: at cfg2 (122 pairs per window, 64 buckets per wavefront) 512-pair chunks make the scoring kernel
// 4 % faster than 256; with 512 buckets per wavefront (k = 11, 12: TBL = 32768) the chunks fill too slowly and 256 stays better
// (cfg3 share: +1.7 % with 512).
#ifndef IPK_CH512_TBL
#define IPK_CH512_TBL 0u         // tuning: a second table size whose chunks hold 512 pairs (32768: DNA k = 11, 12)
#endif
template <uint32_t TBL> constexpr uint32_t chunk_pairs() { return (TBL == 16384 || TBL == IPK_CH512_TBL) ? 512u : 256u; }
inline uint32_t chunk_pairs_rt(uint32_t tbl) { return (tbl == 16384 || tbl == IPK_CH512_TBL) ? 512u : 256u; }
constexpr uint32_t CHUNK_NONE = 0xFFFFFFFFu;
constexpr uint32_t ALLOC_BATCH = 32;         // chunk ids a wavefront draws per global atomic
constexpr uint32_t SUB = 1;                  // open chunks per (wave, bucket); >1 spreads a bucket over lane-interleaved chunks

struct StreamParams {
    const float* logp;
    const float* best;
    const uint32_t* gm_off;        // [G+1] matrices of each group of the batch (CSR)
    const uint32_t* gm_list;       // matrix indices
    uint32_t sites, nwin, tiles_per_mat, S;   // S = segments (workgroups) per group
    float eps;
    uint2* pool;                   // [pool_cap][chunk_pairs<TBL>()] pairs (dense code, score bits)
    uint32_t pool_cap;
    uint32_t* pool_next;           // next free chunk
    unsigned long long* desc;      // [pool_cap] (group * NB + bucket) << 32 | count ; 0 = unused
    uint32_t* pool_ovf;            // set when the pool ran out (the batch is then redone)
    unsigned long long* emitted;
    unsigned long long* ovf_queue;
    uint32_t* ovf_count;
    const uint32_t* mat_slot;      // [n_mats] group slot of a matrix inside the batch (big-list kernel)
    uint32_t flags;                // diagnostics (timing experiments and tests; results are wrong with bits 0, 1, 2, 4 set):
                                   // bit 0 = skip the pool stores, bit 1 = skip the append bookkeeping too (row-per-lane join: no store pass),
                                   // bit 2 = list building only, bit 3 = quad kernel rebases its store window every 8 chunks (tests),
                                   // bit 4 = row-per-lane join: count pass only; (host) bit 5 = no first chunks by position,
                                   // (host) bit 6 = every wait of the call as in round 2 (no device-side counts, no estimated allocations),
                                   // (host) bit 7 = fixed tile ranges, bit 8 = drawn tiles whatever the tile count (tests),
                                   // (host) bit 9 = dense key-major writer with tile-by-tile stores, bit 10 = with line-cut stores, whatever the group count
                                   // (host) bit 11 = 128-KB slices reduced by the workgroup-per-slice kernel instead of the persistent one,
                                   // (host) bit 12 = compressed key-major writer per key block, bit 13 = per run of blocks whatever the group count
    uint32_t pre_chunks;           // row-per-lane quad kernel: chunks [0, pre_chunks) are handed out by position -- wavefront w's bucket b
                                   // starts in chunk w * NB + b -- and pool_next starts at pre_chunks (0: every first chunk is drawn)
    uint32_t* tile_next = nullptr; // quad kernel: [groups] next tile of each group -- its S workgroups DRAW their tiles instead of
                                   // owning a fixed range each (nullptr: fixed ranges); set to S before the launch (tile `seg` is a workgroup's first)
    uint32_t* big_ovf = nullptr;   // set when a window's half list exceeds the big-list kernels' capped capacity (big_capf, DNA k >= 13)
};

// M with floor(t / n) == (t * M) >> 16 for 0 <= t < 128, 1 <= n <= 64   (M = ceil(65536 / n))
__device__ __constant__ uint32_t RCP16[65] = {
    0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096,
    3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2048,
    1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599, 1561, 1525, 1490, 1457, 1425, 1395, 1366,
    1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041, 1024};

// Both halves of the window are joins of two small nodes (sigma^h <= 64 each): DNA k = 8..12.
template <int SIGMA, int K>
struct HalvesDD {
    static constexpr int HL = K / 2, HR = K - K / 2;
    static constexpr int LA = HL / 2, LB = HL - LA, RA = HR / 2, RB = HR - RA;
    static constexpr bool OK = ipow(SIGMA, HL) > 64 && ipow(SIGMA, LB) <= 64 && ipow(SIGMA, RB) <= 64;
    static constexpr uint32_t FLA = ipow(SIGMA, LA), FLB = ipow(SIGMA, LB), FRA = ipow(SIGMA, RA), FRB = ipow(SIGMA, RB);
};
template <int SIGMA, int K, int CAP>
constexpr uint32_t stream_wave_scratch()
{
    using D = HalvesDD<SIGMA, K>;
    if constexpr (D::OK)
        return Geo<SIGMA, D::HL, CAP>::CAPH + Geo<SIGMA, D::HR, CAP>::CAPH + D::FLA + D::FLB + D::FRA + D::FRB;   // + the four child lists
    else
        return wave_scratch_entries<SIGMA, K, CAP>();
}

// List building with the four child nodes evaluated back to back (all their LDS reads in flight
// together) and both half joins sharing one step when they fit 64 candidates each.  Same sets and
// same arithmetic as Node<>::build; only the order of independent work differs.
template <int SIGMA, int K, int CAP>
__device__ __forceinline__ bool build_halves_dd(const WinCtx& c, float eps, uint2* scratch,
                                                const uint2*& Lp, uint32_t& nL, const uint2*& Rp, uint32_t& nR)
{
    using D = HalvesDD<SIGMA, K>;
    constexpr int HL = D::HL, HR = D::HR, LA = D::LA, LB = D::LB, RA = D::RA, RB = D::RB;
    using GL = Geo<SIGMA, HL, CAP>;
    using GR = Geo<SIGMA, HR, CAP>;
    const uint32_t lane = lane_id();
    const float* bs = c.best + c.w;
    const float eps_l = eps - (bs[K] - bs[HL]);                      // pk_compute.cpp:54 at (0, K)
    const float eps_r = eps - (bs[HL] - bs[0]);                      // :55
    const float eps_la = eps_l - (bs[HL] - bs[LA]);                  // :54 at (0, HL)
    const float eps_lb = eps_l - (bs[LA] - bs[0]);                   // :55
    const float eps_ra = eps_r - (bs[K] - bs[HL + RA]);              // :54 at (HL, HR)
    const float eps_rb = eps_r - (bs[HL + RA] - bs[HL]);             // :55
    uint2* lp = scratch;
    uint2* rp = lp + GL::CAPH;
    uint2* la = rp + GR::CAPH;
    uint2* lb = la + D::FLA;
    uint2* ra = lb + D::FLB;
    uint2* rb = ra + D::FRA;
    Lp = lp; Rp = rp; nL = 0; nR = 0;

    float sla = 0.f, slb = 0.f, sra = 0.f, srb = 0.f;
    bool pla = false, plb = false, pra = false, prb = false;
    if (lane < D::FLA) pla = Direct<SIGMA, 0, LA>::eval(c, eps_la, lane, sla);
    if (lane < D::FLB) plb = Direct<SIGMA, LA, LB>::eval(c, eps_lb, lane, slb);
    if (lane < D::FRA) pra = Direct<SIGMA, HL, RA>::eval(c, eps_ra, lane, sra);
    if (lane < D::FRB) prb = Direct<SIGMA, HL + RA, RB>::eval(c, eps_rb, lane, srb);
    const uint64_t mla = __ballot(pla), mlb = __ballot(plb), mra = __ballot(pra), mrb = __ballot(prb);
    if (mla == 0 || mlb == 0 || mra == 0 || mrb == 0) return true;         // an empty half: nothing survives
    if (pla) la[mbcnt(mla)] = make_uint2(lane, __float_as_uint(sla));
    if (plb) lb[mbcnt(mlb)] = make_uint2(lane, __float_as_uint(slb));
    if (pra) ra[mbcnt(mra)] = make_uint2(lane, __float_as_uint(sra));
    if (prb) rb[mbcnt(mrb)] = make_uint2(lane, __float_as_uint(srb));
    const uint32_t nla = (uint32_t)__popcll(mla), nlb = (uint32_t)__popcll(mlb);
    const uint32_t nra = (uint32_t)__popcll(mra), nrb = (uint32_t)__popcll(mrb);
    wave_lds_sync();
    const uint32_t tl = nla * nlb, tr = nra * nrb;
    uint32_t cl, cr;
    if (tl <= 64 && tr <= 64) {
        // one step for both halves
        const uint32_t il = (lane * RCP16[nlb]) >> 16, jl = lane - il * nlb;
        const uint32_t ir = (lane * RCP16[nrb]) >> 16, jr = lane - ir * nrb;
        uint2 a0 = make_uint2(0, 0), b0 = a0, a1 = a0, b1 = a0;
        if (lane < tl) { a0 = la[il]; b0 = lb[jl]; }
        if (lane < tr) { a1 = ra[ir]; b1 = rb[jr]; }
        const float s0 = __uint_as_float(a0.y) + __uint_as_float(b0.y);          // :90
        const float s1 = __uint_as_float(a1.y) + __uint_as_float(b1.y);
        const bool p0 = (lane < tl) && (s0 > eps_l);                             // :91
        const bool p1 = (lane < tr) && (s1 > eps_r);
        const uint64_t m0 = __ballot(p0), m1 = __ballot(p1);
        if (p0) lp[mbcnt(m0)] = make_uint2(a0.x * D::FLB + b0.x, __float_as_uint(s0));
        if (p1) rp[mbcnt(m1)] = make_uint2(a1.x * D::FRB + b1.x, __float_as_uint(s1));
        cl = (uint32_t)__popcll(m0); cr = (uint32_t)__popcll(m1);
    } else {
        cl = join_to_list(la, nla, lb, nlb, eps_l, D::FLB, lp, GL::CAPH);
        if (cl == LIST_OVERFLOW) return false;
        if (cl == 0) return true;
        cr = join_to_list(ra, nra, rb, nrb, eps_r, D::FRB, rp, GR::CAPH);
        if (cr == LIST_OVERFLOW) return false;
    }
    nL = cl; nR = cr;
    wave_lds_sync();
    return true;
}

// Per-wave appender of surviving (code, score) pairs to the pair pool: one open chunk per key bucket.
template <uint32_t TBL, uint32_t NB>
struct Appender {
    static constexpr uint32_t CH = chunk_pairs<TBL>();
    const StreamParams& p;
    uint32_t* cbase;
    uint32_t* cfill;
    uint32_t g;
    uint32_t chunk_next = 0, chunk_end = 0;            // this wave's private range of free chunk ids

    // The chunk of state slot `bb` (= bucket * SUB + sub) is full: close it, open a new one, return its id
    // (CHUNK_NONE if the pool ran out).
    __device__ __forceinline__ uint32_t roll(uint32_t bb)
    {
        const uint32_t lane = lane_id();
        // chunk ids are drawn ALLOC_BATCH at a time: one returning atomic on a single word saturates at
        // ~88 per microsecond chip-wide, far below one per 256 pairs
        if (chunk_next == chunk_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(p.pool_next, ALLOC_BATCH);
            chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            chunk_end = chunk_next + ALLOC_BATCH;
        }
        uint32_t nid = chunk_next++;
        if (nid >= p.pool_cap) { if (lane == 0) atomicOr(p.pool_ovf, 1u); nid = CHUNK_NONE; }
        wave_lds_sync();
        const uint32_t old_id = cbase[bb];
        wave_lds_sync();
        if (lane == 0) {
            if (old_id != CHUNK_NONE) p.desc[old_id] = ((unsigned long long)(g * NB + bb / SUB) << 32) | (unsigned long long)CH;
            cbase[bb] = nid;
            cfill[bb] = cfill[bb] - CH;
        }
        wave_lds_sync();
        return nid;
    }

    // Two candidates per lane (two steps of the final join): slots are reserved with LDS atomics, both
    // reservations in flight together.  (Ballot-ranked variants without same-address atomics were
    // measured 40-55 % slower: the extra 64-bit lane masks push the kernel into SGPR spilling.)
    __device__ __forceinline__ void append2(bool pass0, uint32_t idx0, uint32_t sb0, bool pass1, uint32_t idx1, uint32_t sb1)
    {
        const uint32_t sub = lane_id() & (SUB - 1);
        const uint32_t bk0 = pass0 ? (idx0 / TBL) * SUB + sub : 0u, bk1 = pass1 ? (idx1 / TBL) * SUB + sub : 0u;
        uint32_t slot0 = 0, slot1 = 0, cb0 = CHUNK_NONE, cb1 = CHUNK_NONE;
        if (pass0) slot0 = atomicAdd(&cfill[bk0], 1u);
        if (pass1) slot1 = atomicAdd(&cfill[bk1], 1u);
        if (pass0) cb0 = cbase[bk0];
        if (pass1) cb1 = cbase[bk1];
        const bool st = !(p.flags & 1u);
        if (pass0 && slot0 < CH && cb0 != CHUNK_NONE && st) p.pool[(size_t)cb0 * CH + slot0] = make_uint2(idx0, sb0);
        if (pass1 && slot1 < CH && cb1 != CHUNK_NONE && st) p.pool[(size_t)cb1 * CH + slot1] = make_uint2(idx1, sb1);
        const bool o0 = pass0 && slot0 >= CH, o1 = pass1 && slot1 >= CH;
        uint64_t ovf0 = __ballot(o0), ovf1 = __ballot(o1);
        while (ovf0 | ovf1) {                           // a bucket's chunk filled up: open a new one
            uint32_t bb;
            if (ovf0) bb = (uint32_t)__builtin_amdgcn_readlane((int)bk0, (int)(__ffsll((long long)ovf0) - 1));
            else      bb = (uint32_t)__builtin_amdgcn_readlane((int)bk1, (int)(__ffsll((long long)ovf1) - 1));
            const bool h0 = o0 && bk0 == bb, h1 = o1 && bk1 == bb;
            const uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
            const uint32_t nid = roll(bb);
            if (h0 && nid != CHUNK_NONE && st) p.pool[(size_t)nid * CH + (slot0 - CH)] = make_uint2(idx0, sb0);
            if (h1 && nid != CHUNK_NONE && st) p.pool[(size_t)nid * CH + (slot1 - CH)] = make_uint2(idx1, sb1);
            ovf0 &= ~m0; ovf1 &= ~m1;
        }
    }
};

template <int SIGMA, int K, int CAP, int TW, int NW, uint32_t TBL>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_num_sgpr(80))) void score_stream_kernel(StreamParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t CH = chunk_pairs<TBL>();
    using TG = TileGeo<SIGMA, K, TW>;
    constexpr uint32_t T = ipow(SIGMA, K);
    constexpr uint32_t NB = (T + TBL - 1) / TBL;
    constexpr uint32_t WS = stream_wave_scratch<SIGMA, K, CAP>();
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch_all = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    uint32_t* state_all = reinterpret_cast<uint32_t*>(scratch_all + (size_t)NW * WS);   // per wave: base[NB*SUB], fill[NB*SUB]

    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x / p.S, seg = blockIdx.x - g * p.S;
    const uint32_t m0 = p.gm_off[g], nm = p.gm_off[g + 1] - m0;
    const uint32_t total_tiles = nm * p.tiles_per_mat;
    const uint32_t t_lo = (uint32_t)(((uint64_t)total_tiles * seg) / p.S);
    const uint32_t t_hi = (uint32_t)(((uint64_t)total_tiles * (seg + 1)) / p.S);

    uint2* scratch = scratch_all + (size_t)wave * WS;
    uint32_t* cbase = state_all + (size_t)wave * 2 * NB * SUB;
    uint32_t* cfill = cbase + NB * SUB;
    for (uint32_t b = lane; b < NB * SUB; b += 64) { cbase[b] = CHUNK_NONE; cfill[b] = CH; }
    Appender<TBL, NB> app{p, cbase, cfill, g};
    unsigned long long emitted = 0;                     // per-wave count of scored phylo-k-mers (can pass 2^32 on flat data)
    constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);

    for (uint32_t t = t_lo; t < t_hi; ++t) {
        const uint32_t q = t / p.tiles_per_mat, tile = t - q * p.tiles_per_mat;
        const uint32_t mat = p.gm_list[m0 + q];
        const uint32_t t0 = tile * TW;
        const uint32_t nw = min((uint32_t)TW, p.nwin - t0);
        const uint32_t ncol = nw + K - 1;
        __syncthreads();                                   // previous tile fully consumed
        {
            const float4* src = reinterpret_cast<const float4*>(p.logp + ((size_t)mat * p.sites + t0) * SIGMA);
            float4* dst = reinterpret_cast<float4*>(cols);
            const uint32_t n4 = ncol * (SIGMA / 4);
            for (uint32_t i = threadIdx.x; i < n4; i += NW * 64) dst[i] = src[i];
            const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + t0;
            for (uint32_t i = threadIdx.x; i <= ncol; i += NW * 64) best[i] = bsrc[i];
        }
        __syncthreads();
        for (uint32_t w = wave; w < nw; w += NW) {
            WinCtx c{cols, best, w};
            const uint2 *L, *R;
            uint32_t nL, nR;
            static_assert(!Geo<SIGMA, K, CAP>::DIRECT, "stream variant needs k with sigma^k > 64");
            bool ok;
            if constexpr (HalvesDD<SIGMA, K>::OK) ok = build_halves_dd<SIGMA, K, CAP>(c, p.eps, scratch, L, nL, R, nR);
            else ok = build_halves<SIGMA, K, CAP>(c, p.eps, scratch, L, nL, R, nR);
            if (!ok) {
                if (lane == 0) {
                    const uint32_t qi = atomicAdd(p.ovf_count, 1u);
                    p.ovf_queue[qi] = ((unsigned long long)mat << 32) | (unsigned long long)(t0 + w);
                }
                continue;
            }
            if (nL == 0 || nR == 0) continue;
            if (p.flags & 4u) { emitted += nL + nR; continue; }                 // diagnostics: list building only
            uint32_t cnt = 0;
            if (nR < 64) {
                // flattened candidate space, two steps (128 candidates) per trip; (i, j) of a candidate
                // from a scalar (bi, bj) of the step's first candidate plus one multiply-shift per lane
                const uint32_t total = nL * nR;
                const uint32_t M = RCP16[nR];
                const uint32_t q64 = (64u * M) >> 16, r64 = 64u - q64 * nR;
                uint32_t bi = 0, bj = 0;
                for (uint32_t base = 0; base < total; base += 128) {
                    uint32_t bi1 = bi + q64, bj1 = bj + r64;
                    if (bj1 >= nR) { bj1 -= nR; ++bi1; }
                    const uint32_t ta = bj + lane, tb = bj1 + lane;
                    const uint32_t qa = (ta * M) >> 16, qb = (tb * M) >> 16;
                    const bool va = base + lane < total, vb = base + 64 + lane < total;
                    uint2 a0 = make_uint2(0, 0), b0 = a0, a1 = a0, b1 = a0;
                    if (va) { a0 = L[bi + qa]; b0 = R[ta - qa * nR]; }
                    if (vb) { a1 = L[bi1 + qb]; b1 = R[tb - qb * nR]; }
                    const float s0 = __uint_as_float(a0.y) + __uint_as_float(b0.y);      // pk_compute.cpp:90
                    const float s1 = __uint_as_float(a1.y) + __uint_as_float(b1.y);
                    const bool p0 = va && (s0 > p.eps), p1 = vb && (s1 > p.eps);          // :91
                    cnt += (uint32_t)__popcll(__ballot(p0)) + (uint32_t)__popcll(__ballot(p1));
                    if (!(p.flags & 2u))
                        app.append2(p0, a0.x * mulR + b0.x, __float_as_uint(s0), p1, a1.x * mulR + b1.x, __float_as_uint(s1));
                    bi = bi1 + q64; bj = bj1 + r64;
                    if (bj >= nR) { bj -= nR; ++bi; }
                }
            } else {
                for (uint32_t i = 0; i < nL; ++i) {
                    const uint2 a = L[i];
                    for (uint32_t jb = 0; jb < nR; jb += 128) {
                        const uint32_t ja = jb + lane, jc = jb + 64 + lane;
                        const bool va = ja < nR, vb = jc < nR;
                        uint2 b0 = make_uint2(0, 0), b1 = b0;
                        if (va) b0 = R[ja];
                        if (vb) b1 = R[jc];
                        const float s0 = __uint_as_float(a.y) + __uint_as_float(b0.y);
                        const float s1 = __uint_as_float(a.y) + __uint_as_float(b1.y);
                        const bool p0 = va && (s0 > p.eps), p1 = vb && (s1 > p.eps);
                        cnt += (uint32_t)__popcll(__ballot(p0)) + (uint32_t)__popcll(__ballot(p1));
                        if (!(p.flags & 2u))
                            app.append2(p0, a.x * mulR + b0.x, __float_as_uint(s0), p1, a.x * mulR + b1.x, __float_as_uint(s1));
                    }
                }
            }
            emitted += cnt;
        }
    }
    // close this wave's open chunks
    wave_lds_sync();
    for (uint32_t b = lane; b < NB * SUB; b += 64) {
        const uint32_t id = cbase[b];
        if (id != CHUNK_NONE) p.desc[id] = ((unsigned long long)(g * NB + b / SUB) << 32) | (unsigned long long)min(cfill[b], CH);
    }
    if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);
}

// Big-list windows of the stream variant: same cooperative shape as score_overflow_kernel (wave 0 builds
// the half lists at worst-case capacity, all waves share the final cross product), but the survivors are
// appended to the pair pool like everything else, so they are max-reduced in LDS by pass 2 instead of by
// global atomics.  Runs between pass 1 and the chunk index.  A wave keeps its chunks open while
// consecutive queue entries belong to the same group.
template <int SIGMA, int K, uint32_t TBL>
__global__ __launch_bounds__(OVF_NW * 64) void score_overflow_stream_kernel(StreamParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t CH = chunk_pairs<TBL>();
    constexpr int CAPF = big_capf<SIGMA, K>();
    constexpr uint32_t T = ipow(SIGMA, K);
    constexpr uint32_t NB = (T + TBL - 1) / TBL;
    using TG = TileGeo<SIGMA, K, 1>;
    __shared__ uint32_t sh_n[2];
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    uint32_t* state_all = reinterpret_cast<uint32_t*>(scratch + wave_scratch_entries<SIGMA, K, CAPF>());
    const uint32_t n = *p.ovf_count;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t* cbase = state_all + (size_t)wave * 2 * NB * SUB;
    uint32_t* cfill = cbase + NB * SUB;
    for (uint32_t b = lane; b < NB * SUB; b += 64) { cbase[b] = CHUNK_NONE; cfill[b] = CH; }
    Appender<TBL, NB> app{p, cbase, cfill, 0u};
    bool have_group = false;
    unsigned long long emitted = 0;
    constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
    auto close_chunks = [&]() {
        wave_lds_sync();
        for (uint32_t b = lane; b < NB * SUB; b += 64) {
            const uint32_t id = cbase[b];
            if (id != CHUNK_NONE) p.desc[id] = ((unsigned long long)(app.g * NB + b / SUB) << 32) | (unsigned long long)min(cfill[b], CH);
            cbase[b] = CHUNK_NONE; cfill[b] = CH;
        }
        wave_lds_sync();
    };
    // the queue arrives sorted by group (ovf_group_keys_kernel + radix sort); each workgroup takes one contiguous
    // block of it, so a wave's open chunks are closed only at the few group boundaries inside its block
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t q_lo = min(n, blockIdx.x * per), q_hi = min(n, q_lo + per);
    for (uint32_t q = q_lo; q < q_hi; ++q) {
        const unsigned long long e = p.ovf_queue[q];
        const uint32_t g = (uint32_t)(e >> 42), mat = (uint32_t)(e >> 21) & 0x1FFFFFu, start = (uint32_t)e & 0x1FFFFFu;
        if (have_group && g != app.g) close_chunks();
        app.g = g; have_group = true;
        __syncthreads();                                                       // previous window's lists consumed
        const float* src = p.logp + ((size_t)mat * p.sites + start) * SIGMA;
        for (uint32_t i = threadIdx.x; i < K * SIGMA; i += OVF_NW * 64) cols[i] = src[i];
        const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + start;
        for (uint32_t i = threadIdx.x; i <= K; i += OVF_NW * 64) best[i] = bsrc[i];
        __syncthreads();
        const uint2 *L = scratch, *R = scratch + Geo<SIGMA, K / 2, CAPF>::CAPH;
        if (wave == 0) {
            WinCtx c{cols, best, 0};
            uint32_t nL = 0, nR = 0;
            const bool fits = build_halves<SIGMA, K, CAPF>(c, p.eps, scratch, L, nL, R, nR);      // (cannot overflow at full capacity)
            if (!fits) { nL = 0; nR = 0; if (lane == 0 && p.big_ovf) atomicOr(p.big_ovf, 1u); }   // capped lists (big_capf): the call fails
            if (lane == 0) { sh_n[0] = nL; sh_n[1] = nR; }
        }
        __syncthreads();
        const uint32_t nL = sh_n[0], nR = sh_n[1];
        if (nL == 0 || nR == 0) continue;
        uint32_t cnt = 0;
        for (uint32_t i = wave; i < nL; i += OVF_NW) {                         // rows of L dealt to the waves
            const uint2 a = L[i];
            for (uint32_t jb = 0; jb < nR; jb += 128) {
                const uint32_t ja = jb + lane, jc = jb + 64 + lane;
                const bool va = ja < nR, vb = jc < nR;
                uint2 b0 = make_uint2(0, 0), b1 = b0;
                if (va) b0 = R[ja];
                if (vb) b1 = R[jc];
                const float s0 = __uint_as_float(a.y) + __uint_as_float(b0.y);   // pk_compute.cpp:90
                const float s1 = __uint_as_float(a.y) + __uint_as_float(b1.y);
                const bool p0 = va && (s0 > p.eps), p1 = vb && (s1 > p.eps);      // :91
                cnt += (uint32_t)__popcll(__ballot(p0)) + (uint32_t)__popcll(__ballot(p1));
                app.append2(p0, a.x * mulR + b0.x, __float_as_uint(s0), p1, a.x * mulR + b1.x, __float_as_uint(s1));
            }
        }
        emitted += cnt;
    }
    if (have_group) close_chunks();
    if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);
}

// (mat << 32 | start) -> (group << 42 | mat << 21 | start): sortable by group (host checks the field widths)
__global__ __launch_bounds__(256) void ovf_group_keys_kernel(const unsigned long long* __restrict__ queue, uint32_t n,
                                                             const uint32_t* __restrict__ mat_slot,
                                                             unsigned long long* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long e = queue[i];
    const uint32_t mat = (uint32_t)(e >> 32), start = (uint32_t)e;
    out[i] = ((unsigned long long)mat_slot[mat] << 42) | ((unsigned long long)mat << 21) | (unsigned long long)start;
}

// chunk index: how many chunks each (group, bucket) has; then chunk ids grouped by (group, bucket)
// `pairs` (optional): the pool's pair total is added there -- the quad kernel leaves the count of scored phylo-k-mers
// (the reference's `count`, db_builder.cpp:664) to this pass instead of counting per step.  Grid-stride over the
// descriptors with at most a few thousand workgroups: one atomic per workgroup on the total (a single word takes
// ~88 returning atomics per microsecond chip-wide, so one per wavefront of descriptors would cost milliseconds).
// (n_dev: the number of chunk ids drawn, still on the device -- a call that does not wait for it on the host passes the pool's
//  capacity as n and the counter here; both kernels then stop at min(*n_dev, n))
__global__ __launch_bounds__(256) void chunk_hist_kernel(const unsigned long long* __restrict__ desc, uint32_t n,
                                                         uint32_t* __restrict__ cnt, unsigned long long* __restrict__ pairs,
                                                         uint32_t* __restrict__ gb_pairs /* optional: pairs per (group, bucket) */,
                                                         const uint32_t* __restrict__ n_dev = nullptr)
{
    __shared__ unsigned long long wsum[4];
    unsigned long long c = 0;
    if (n_dev) n = min(n, *n_dev);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned long long d = desc[i];
        if ((uint32_t)d != 0u) {
            atomicAdd(&cnt[(uint32_t)(d >> 32)], 1u);
            if (gb_pairs) atomicAdd(&gb_pairs[(uint32_t)(d >> 32)], (uint32_t)d);
        }
        c += (uint32_t)d;
    }
    if (pairs) {
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane_id() == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (t) atomicAdd(pairs, t);
        }
    }
}
__global__ __launch_bounds__(256) void chunk_scatter_kernel(const unsigned long long* __restrict__ desc, uint32_t n,
                                                            const uint64_t* __restrict__ off, uint32_t* __restrict__ cur,
                                                            uint2* __restrict__ list, const uint32_t* __restrict__ n_dev = nullptr)
{
    if (n_dev) n = min(n, *n_dev);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned long long d = desc[i];
        if ((uint32_t)d == 0u) continue;
        const uint32_t gb = (uint32_t)(d >> 32);
        list[off[gb] + atomicAdd(&cur[gb], 1u)] = make_uint2(i, (uint32_t)d);      // (chunk id, pair count)
    }
}

// The per-batch counters of ctx->small in one launch (they were five 4-us fills): scored k-mers @0, big-list queue length @16,
// chunk ids drawn @32 (starting behind the chunks handed out by position), pool-exhausted flag @36, spare counter @48.
__global__ void small_reset_kernel(uint32_t* __restrict__ small, uint32_t pre_chunks, uint32_t* __restrict__ tile_next, uint32_t n_groups,
                                   uint32_t first_tile)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 14) small[i] = i == 8 ? pre_chunks : 0u;
    if (tile_next && i < n_groups) tile_next[i] = first_tile;   // (the quad kernel's per-group tile counters ride along: tiles [0, S) go by position)
}

// Occupancy bits of a finished LDS table slice (slots key0 .. key0 + nslots of one group; key0 a multiple of 64):
// bit x % 32 of word x / 32 says table[x] != 0.  km_count sums these bits instead of re-reading the dense tables.
__device__ __forceinline__ void store_slice_mask(const uint32_t* tab, uint32_t nslots, uint32_t* mask_words, uint32_t nthreads)
{
    // every wavefront takes a CONTIGUOUS range of 64-slot blocks and lane j keeps block j's bits, so that they leave as one coalesced
    // 8-byte-per-lane store per 64 blocks (one lane-0 store of two words per block was 250 partly written lines per slice)
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nwv = nthreads >> 6;
    const uint32_t nblk = (nslots + 63) / 64;
    const uint32_t bpw = (nblk + nwv - 1) / nwv;
    const uint32_t b_lo = min(nblk, wave * bpw), b_hi = min(nblk, b_lo + bpw);
    for (uint32_t b0 = b_lo; b0 < b_hi; b0 += 64) {
        const uint32_t nb = min(64u, b_hi - b0);
        uint64_t mine = 0;
#pragma unroll 4
        for (uint32_t j = 0; j < nb; ++j) {
            const uint32_t c = (b0 + j) * 64;
            const uint32_t v = (c + lane < nslots) ? tab[c + lane] : 0u;
            const uint64_t m = __ballot(v != 0u);
            if (lane == j) mine = m;
        }
        // rows are padded to whole 64-slot blocks (mask_words = 2 * ceil(T / 64))
        if (lane < nb) *reinterpret_cast<uint2*>(mask_words + 2 * (b0 + lane)) = make_uint2((uint32_t)mine, (uint32_t)(mine >> 32));
    }
}

// pass 2: one workgroup per (group, bucket): LDS max-reduce of the bucket's chunks, then the table slice.
// Each wave takes two chunks per trip and issues all of their pair loads (8 x 512 B) before the first
// LDS atomic, so ~4 KiB per wave are in flight.
// Room for the values of a compressed slice (below): a (group, bucket) holds at most min(its pairs, its slots) distinct keys;
// sizes in 8-byte units (the CompTable offsets address a uint2 array).
__global__ __launch_bounds__(256) void comp_slice_room_kernel(const uint32_t* __restrict__ gb_pairs, uint32_t n_gb, uint32_t NB, uint64_t T,
                                                              uint32_t TBL, uint32_t* __restrict__ room)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_gb) return;
    const uint32_t b = i % NB;
    const uint32_t nslots = (uint32_t)min((uint64_t)TBL, T - (uint64_t)b * TBL);
    room[i] = (min(gb_pairs[i], nslots) + 1u) / 2u;
}

// The compressed form of a finished table slice (comp_table.hpp): occupancy bits, rank of every 64-slot block, the non-empty
// slots' score codes in slot order at `vals`, and the blocks' value addresses.  `tab` = the slice in LDS, PADDED to
// comp_padded_slots<TBL, NT>() slots (every wavefront owns exactly BPW whole blocks; the slots past the slice are zero and stay zero:
// no key maps there), then NT / 64 + 1 words; every thread of the workgroup calls it after the barrier that ends the reduction.
//
// Round 4: this epilogue was most of the reduce kernels' vector instructions (r04_*_backhalf_sq.json: 960 VALU per wavefront and
// slice at AA k = 6, of which 775 here -- 24 per 64-slot block -- with the SIMDs' vector pipes 83 % busy), so it is written for
// instruction count: no bounds tests (padding), block addresses as immediate LDS offsets, the per-block rank and bits gathered with
// v_writelane (one instruction per value, no compare), the value stores as scalar base + 32-bit lane offset.  ZERO: the slots are
// cleared on the way out (the persistent kernel's next slice starts from a clean table; needs a barrier before the next atomics).
#ifndef IPK_RB_ABL
#define IPK_RB_ABL 0             // timing experiments (results wrong, every address stays valid): 1 = no pair loads, 2 = no LDS atomics, 4 = no value stores in the compress epilogue
#endif
#ifndef IPK_COMPACT
#define IPK_COMPACT 0            // compress epilogue, 1: values compacted in LDS per wavefront and copied out as full-wave stores -- measured SLOWER than
                                 // one partial store per 64-slot block (cfg3 share 3.26 against 3.03 ms, cfg4 14.0 against 13.8): the runs merge in L2 anyway
#endif
// v_writelane_b32 with the lane as an immediate: lane LANE of v := the wave-uniform value s (no builtin in this compiler)
template <uint32_t LANE>
__device__ __forceinline__ void writelane_imm(uint32_t& v, uint32_t s)       // (s_nop: see writelane64_imm; s may come from v_readfirstlane)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(LANE));
}
// both halves of a 64-bit wave-uniform value (a ballot: written by a VALU compare).  gfx950 needs two wait states between a VALU
// write of an SGPR and a VALU read of it; the compiler pads its own instructions but cannot see into an asm statement -- without
// the s_nop the v_writelane right behind the v_cmp picked up the PREVIOUS block's bits (keys 64 slots off).
template <uint32_t LANE>
__device__ __forceinline__ void writelane64_imm(uint32_t& vlo, uint32_t& vhi, uint64_t s)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(vlo), "+v"(vhi) : "s"((uint32_t)s), "s"((uint32_t)(s >> 32)), "n"(LANE));
}
template <uint32_t N, class F, uint32_t... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<uint32_t, I...>) { (f(std::integral_constant<uint32_t, I>{}), ...); }
template <uint32_t N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<uint32_t, N>{}); }

template <uint32_t TBL, int NT> constexpr uint32_t comp_bpw() { return ((TBL + 63) / 64 + NT / 64 - 1) / (NT / 64); }
template <uint32_t TBL, int NT> constexpr uint32_t comp_padded_slots() { return comp_bpw<TBL, NT>() * (NT / 64) * 64; }
// REREAD: the blocks are read from LDS a second time for the write-out instead of being kept in registers (the persistent kernel has
// 64 registers of pair loads in flight through here).
template <uint32_t TBL, int NT, bool ZERO = false, bool REREAD = false>
__device__ __forceinline__ void compress_slice(uint32_t* tab, uint32_t nslots, uint32_t* vals, uint32_t* mrow, uint32_t* rrow,
                                               uint64_t* arow, uint32_t* ucnt_gb)
{
    constexpr uint32_t NWV = NT / 64, BPW = comp_bpw<TBL, NT>(), PAD = comp_padded_slots<TBL, NT>();
    static_assert(BPW <= 64, "one lane per block of the wave");
    uint32_t* wtot = tab + PAD;                                   // [NWV + 1]
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nblk = (nslots + 63) / 64;
    uint32_t* wb = tab + wave * (BPW * 64) + lane;                // block j of this wavefront: wb[j * 64]
    // the wave's blocks in registers (all LDS reads in flight together), then counting and writing run on registers
    uint32_t vr[REREAD ? 1 : BPW];
    uint32_t mine = 0;
    if constexpr (REREAD) {
#pragma unroll
        for (uint32_t j = 0; j < BPW; ++j) mine += (uint32_t)__popcll(__ballot(wb[j * 64] != 0u));
    } else {
#pragma unroll
        for (uint32_t j = 0; j < BPW; ++j) vr[j] = wb[j * 64];
#pragma unroll
        for (uint32_t j = 0; j < BPW; ++j) mine += (uint32_t)__popcll(__ballot(vr[j] != 0u));
    }
    if (lane == 0) wtot[wave] = mine;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < NWV; ++w) { const uint32_t t = wtot[w]; if (w < wave) base += t; all += t; }
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    // lane j collects block j's rank and bits; they leave as coalesced stores after the loop
    uint32_t my_base = 0, my_lo = 0, my_hi = 0;
    char* vbytes = reinterpret_cast<char*>(vals);
    if constexpr (IPK_COMPACT != 0) {
        // The wavefront's values are compacted IN PLACE at the head of its own table region (block j's values land at or below
        // block j's slots, which are in registers or already read), then leave as full-wave stores of 64 consecutive values --
        // r04 ablation: the per-block stores (a run of ~20 values at 4-byte alignment each) were 0.5 of 3.0 ms at a cfg3 share
        // and 2.75 of 13.5 ms at cfg4.  No barrier: every wavefront stays inside its own region.
        uint32_t* wreg = tab + wave * (BPW * 64);
        uint32_t lp = 0;
        static_for<BPW>([&](auto J) {
            constexpr uint32_t j = decltype(J)::value;
            uint32_t v;
            if constexpr (REREAD) v = wb[j * 64]; else v = vr[j];
            const uint64_t m = __ballot(v != 0u);
            writelane_imm<j>(my_base, base + lp);
            writelane64_imm<j>(my_lo, my_hi, m);
            if (v != 0u) wreg[lp + mbcnt(m)] = v;
            lp += (uint32_t)__popcll(m);
        });
        uint32_t* dst = vals + base;
        for (uint32_t t = lane; t < lp; t += 64) {
            const uint32_t v = wreg[t];
            if (!(IPK_RB_ABL & 4) || v == 0xFFFFFFFEu) dst[t] = v;
        }
        if constexpr (ZERO) {
            // (everything the wavefront's region may hold: compacted values at the head, blocks not yet overwritten behind them)
#pragma unroll
            for (uint32_t j = 0; j < BPW; ++j) wb[j * 64] = 0u;
        }
    } else {
    static_for<BPW>([&](auto J) {
        constexpr uint32_t j = decltype(J)::value;
        uint32_t v;
        if constexpr (REREAD) v = wb[j * 64]; else v = vr[j];
        const uint64_t m = __ballot(v != 0u);
        writelane_imm<j>(my_base, base);
        writelane64_imm<j>(my_lo, my_hi, m);
        if ((IPK_RB_ABL & 4) ? v == 0xFFFFFFFEu : v != 0u) *reinterpret_cast<uint32_t*>(vbytes + ((base + mbcnt(m)) << 2)) = v;
        if constexpr (ZERO) wb[j * 64] = 0u;
        base += (uint32_t)__popcll(m);
    });
    }
    const uint32_t blk = wave * BPW + lane;
    if (lane < BPW && blk < nblk) {
        rrow[blk] = my_base;
        *reinterpret_cast<uint2*>(mrow + 2 * blk) = make_uint2(my_lo, my_hi);
        arow[blk] = (uint64_t)(vals + my_base);
    }
    if (threadIdx.x == 0) *ucnt_gb = all;
}

// COMPRESS: the slice leaves in the compressed form instead of as TBL dense slots -- for key spaces the scored k-mers fill
// sparsely (DNA k = 12: 38 % of a group's 16.8 M slots at cfg3) the dense tables are the larger part of what pass 2 writes and
// the key-major writer reads.  The values of (group, bucket) go to cvals + coff[group * NB + bucket] (8-byte units; room for
// min(pairs, slots) values each, comp_slice_room_kernel).
#ifndef IPK_RB_CPT
#define IPK_RB_CPT 2
#endif
#ifndef IPK_RB_NT
#define IPK_RB_NT 0              // 1: the reduce kernels read the pair pool with non-temporal loads (read once, never again)
#endif
__device__ __forceinline__ uint2 pool_load(const uint2* p)
{
#if IPK_RB_NT
    const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
#else
    return *p;
#endif
}

template <uint32_t TBL, int NT, bool COMPRESS = false>
__global__ __launch_bounds__(NT) void reduce_buckets_kernel(const uint2* __restrict__ pool,
                                                           const uint64_t* __restrict__ off, const uint2* __restrict__ list,
                                                           uint32_t NB, uint64_t T, uint32_t* __restrict__ table,
                                                           uint32_t* __restrict__ mask, uint64_t mask_words,
                                                           uint2* __restrict__ cvals, const uint64_t* __restrict__ coff,
                                                           uint32_t* __restrict__ rank, uint64_t* __restrict__ vaddr,
                                                           uint32_t* __restrict__ ucnt)
{
    constexpr uint32_t CH = chunk_pairs<TBL>();
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
    constexpr uint32_t NWV = NT / 64;
    const uint32_t gb = blockIdx.x;
    const uint32_t g = gb / NB, b = gb - g * NB;
    const uint64_t key0 = (uint64_t)b * TBL;
    const uint32_t nslots = (uint32_t)min((uint64_t)TBL, T - key0);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t c0 = off[gb], c1 = off[gb + 1];
    // first chunk descriptors are requested before the table is cleared
    constexpr int CPT = IPK_RB_CPT;                         // chunks per wavefront and trip: all their pair loads are issued before the first LDS atomic
    uint64_t ci = c0 + wave;
    uint2 e[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) { e[c] = make_uint2(0, 0); if (ci + (uint64_t)c * NWV < c1) e[c] = list[ci + (uint64_t)c * NWV]; }
    constexpr uint32_t CLR = COMPRESS ? comp_padded_slots<TBL, NT>() : TBL;      // (the compress epilogue reads whole blocks per wavefront)
    for (uint32_t i = threadIdx.x; i < CLR / 4; i += NT) reinterpret_cast<uint4*>(tab)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t k0 = (uint32_t)key0;
    while (ci < c1) {
        constexpr int PER = CH / 64;                        // loads per lane and chunk
        uint2 v[CPT * PER];
        uint32_t n[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            n[c] = (ci + (uint64_t)c * NWV < c1) ? e[c].y : 0u;
            const uint2* s = pool + (size_t)e[c].x * CH;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                v[c * PER + j] = make_uint2(0, 0);
                if constexpr (IPK_RB_ABL & 1) { if (lane + 64 * j < n[c]) v[c * PER + j] = make_uint2(k0 + ((lane * 97u + j * 13u + (uint32_t)ci) & (TBL - 1)), lane); }
                else if (lane + 64 * j < n[c]) v[c * PER + j] = pool_load(s + lane + 64 * j);
            }
        }
        ci += (uint64_t)CPT * NWV;
#pragma unroll
        for (int c = 0; c < CPT; ++c) if (ci + (uint64_t)c * NWV < c1) e[c] = list[ci + (uint64_t)c * NWV];
#pragma unroll
        for (int c = 0; c < CPT; ++c)
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if constexpr (IPK_RB_ABL & 2) { if (v[c * PER + j].x == 0xFFFFFFFFu) tab[0] = v[c * PER + j].y; }
                else if (lane + 64 * j < n[c]) atomicMax(&tab[v[c * PER + j].x - k0], enc_score_bits(v[c * PER + j].y));
    }
    __syncthreads();
    if constexpr (COMPRESS) {
        compress_slice<TBL, NT>(tab, nslots, reinterpret_cast<uint32_t*>(cvals + coff[gb]), mask + (size_t)g * mask_words + (key0 >> 5),
                                rank + (size_t)g * (mask_words / 2) + (key0 >> 6), vaddr + (size_t)g * (mask_words / 2) + (key0 >> 6), ucnt + gb);
        return;
    }
    uint32_t* dst = table + (size_t)g * T + key0;
    if ((nslots & 3u) == 0 && ((((size_t)g * T + key0) & 3u) == 0)) {
        for (uint32_t i = threadIdx.x; i < nslots / 4; i += NT) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<uint4*>(tab)[i];
    } else {
        for (uint32_t i = threadIdx.x; i < nslots; i += NT) dst[i] = tab[i];
    }
    if (mask) store_slice_mask(tab, nslots, mask + (size_t)g * mask_words + (key0 >> 5), NT);
}


// =================================================================================================
// Exact-partition variant ("xp"): for key spaces with thousands of buckets per group (AA k=6: 2000) the
// per-wave open chunks of the stream variant do not fit LDS.  Scoring a window is cheap next to a global
// atomic per scored phylo-k-mer, so the windows are scored TWICE:
//   count  score_xp_kernel<.., false>: pairs per (group, bucket, segment) in workgroup-shared LDS counters
//   scan   exclusive offsets, laid out (group, bucket, segment) so a (group, bucket) range is contiguous
//   write  score_xp_kernel<.., true>: the same windows again; pairs go to their exact pool positions
//   reduce reduce_ranges_kernel: one workgroup per (group, bucket), LDS max-reduce, dense table slice
// The final cross product runs one L entry ("row") at a time: a row's pairs share the high code digits
// a.x * mulR, hence (TBL a multiple of mulR) one bucket, and land in one contiguous run of the pool.
// Counting runs with one row per LANE (64 bucket reservations per LDS atomic instruction).  The write
// pass sorts R by score first, so a row's passing pairs are a prefix of it (length by binary search,
// equal to the count pass' number) and leave as one dense run, the lanes over the prefix.  Same sets, same float operation per pair as for_each_pair/join
// (pk_compute.cpp:90-91).
// =================================================================================================
// rank[c] += #{i < n : key(R[i]) > kj[c]} for the lane's NCH keys; key = (score code << 32) | ~position.  The lanes already hold
// every key (entry j = ch * 64 + lane in kj[ch]): entry i's key is broadcast from there with two v_readlane into scalar registers
// and compared with the lane's keys as a scalar operand -- no LDS read, no re-encoding of a wave-uniform value in every lane, no
// padding entries.  NCH is a compile-time count so the body is branch-free.
template <int NCH, int RC>
__device__ __forceinline__ void xp_rank_by_counting(uint32_t n, const unsigned long long (&kj)[RC], uint32_t (&rank)[RC])
{
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        const uint32_t lo = (uint32_t)kj[ci], hi = (uint32_t)(kj[ci] >> 32);
        const uint32_t m = min(64u, n - min(n, (uint32_t)ci * 64u));           // entries of chunk ci
        for (uint32_t l = 0; l < m; ++l) {
            const unsigned long long ki = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, (int)l) << 32) |
                                          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lo, (int)l);
#pragma unroll
            for (int c = 0; c < NCH; ++c) rank[c] += (ki > kj[c]) ? 1u : 0u;
        }
    }
}

// Bitonic sort, descending, of NCH * 64 64-bit keys held one per (register c, lane): element e = c * 64 + lane; afterwards
// kj[c] of lane l is the key of rank c * 64 + l.  Strides below 64 exchange across lanes (ds_bpermute), strides of 64 and more
// between a lane's own registers.  Keys are distinct (the position is part of them), padding keys are 0 and end up last.
// n = 135 (the mean at AA k=6, four registers): 33 lane stages of 4 x ~7 instructions + 3 register stages against 135 x (2 readlanes
// + 3 64-bit compares) for ranking by counting -- and no O(n^2).
#ifndef IPK_SORT_DPP
#define IPK_SORT_DPP 1
#endif
template <int NCH, int RC>
__device__ __forceinline__ void xp_sort_desc(unsigned long long (&kj)[RC])
{
    static_assert(NCH >= 1 && NCH <= RC && (NCH & (NCH - 1)) == 0, "a power of two of 64-key registers");
    const uint32_t lane = lane_id();
#pragma unroll
    for (int size = 2; size <= NCH * 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            if (stride >= 64) {
                constexpr int dummy = 0; (void)dummy;
                const int cs = stride >> 6;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if ((c & cs) != 0) continue;
                    // block of `size` elements this pair lies in: descending where (e & size) == 0  (e's lane bits do not reach size >= 128)
                    const bool desc = ((c * 64) & size) == 0 || size == NCH * 64;
                    const unsigned long long a = kj[c], b = kj[c | cs];
                    const bool sw = desc ? (a < b) : (a > b);
                    kj[c] = sw ? b : a;
                    kj[c | cs] = sw ? a : b;
                }
            } else {
                const int addr = (int)((lane ^ (uint32_t)stride) << 2);
                const bool lower = (lane & (uint32_t)stride) == 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const unsigned long long a = kj[c];
                    uint32_t olo, ohi;
                    if constexpr (IPK_SORT_DPP != 0) {
                        // partners one or two lanes away sit in the same quad: a DPP move (no LDS round trip) -- 15 of the 33 lane stages
                        // of a 256-key sort
                        if (stride == 1) {
                            olo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)a, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
                            ohi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(a >> 32), 0xB1, 0xF, 0xF, true);
                        } else if (stride == 2) {
                            olo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)a, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
                            ohi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(a >> 32), 0x4E, 0xF, 0xF, true);
                        } else {
                            olo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)a);
                            ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(a >> 32));
                        }
                    } else {
                        olo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)a);
                        ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(a >> 32));
                    }
                    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
                    const uint32_t e = (uint32_t)c * 64u + lane;
                    const bool desc = (e & (uint32_t)size) == 0 || size == NCH * 64;
                    const bool keep_max = lower == desc;                 // the lower index of a descending pair keeps the larger key
                    const bool take = keep_max ? (o > a) : (o < a);
                    kj[c] = take ? o : a;
                }
            }
        }
    }
}

struct XpParams {
    StreamParams s;                // pool_next / desc / pool_ovf unused
    uint32_t* cnt;                 // [(group * NB + bucket) * stride + segment]   (count pass output)
    const uint64_t* off;           // exclusive scan of cnt                        (write pass input)
    uint32_t stride;               // S + 1: segment S of every (group, bucket) belongs to the big-list windows
    uint32_t* ovcur;               // [group * NB + bucket] pairs the big-list write pass has placed so far
    const uint32_t* start;         // [(group * S + segment) * NB + bucket] where the unit's range of the bucket starts, relative to
                                   // off[group * NB * stride] (write pass: a workgroup's NB cursors as one contiguous load)
};

// start[(g * S + seg) * NB + b] = off[(g * NB + b) * stride + seg] - off[g * NB * stride]: the scan's offsets of one workgroup's
// buckets are `stride` elements apart (a (group, bucket) range is contiguous over the segments); the write pass wants them side by
// side -- one global load per row of L from that strided array was 13 GB of line fetches and a dependent load per row block.
__global__ __launch_bounds__(256) void xp_unit_starts_kernel(const uint64_t* __restrict__ off, uint32_t NB, uint32_t S, uint32_t stride,
                                                             uint64_t n, uint32_t* __restrict__ start, uint32_t* __restrict__ too_big)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;       // = (g * S + seg) * NB + b
    if (i >= n) return;
    const uint32_t b = (uint32_t)(i % NB);
    const uint64_t u = i / NB;
    const uint32_t seg = (uint32_t)(u % S);
    const uint64_t g = u / S;
    const uint64_t rel = off[(g * NB + b) * stride + seg] - off[g * NB * stride];
    if (rel >> 32) atomicOr(too_big, 1u);                              // a group with 2^32 pairs or more (the host fails the call)
    start[i] = (uint32_t)rel;
}

template <int SIGMA, int K, int CAP, int TW, int NW, uint32_t TBL, bool WRITE>
__global__ __launch_bounds__(NW * 64) void score_xp_kernel(XpParams xp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const StreamParams& p = xp.s;
    using TG = TileGeo<SIGMA, K, TW>;
    constexpr uint32_t T = ipow(SIGMA, K);
    constexpr uint32_t NB = (T + TBL - 1) / TBL;
    constexpr uint32_t WS = stream_wave_scratch<SIGMA, K, CAP>();
    constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
    static_assert(TBL % mulR == 0, "a row of the final join must stay inside one bucket");
    constexpr uint32_t RPB = TBL / mulR;                    // rows (L codes) per bucket
    using Cursor = uint32_t;                                // count: pairs of the bucket; write: pairs placed so far
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch_all = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    Cursor* cur = reinterpret_cast<Cursor*>(scratch_all + (size_t)NW * WS);     // workgroup-shared, one per bucket

    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t lane8 = lane * 8u;
    const uint32_t g = blockIdx.x / p.S, seg = blockIdx.x - g * p.S;
    const uint32_t m0 = p.gm_off[g], nm = p.gm_off[g + 1] - m0;
    const uint32_t total_tiles = nm * p.tiles_per_mat;
    const uint32_t t_lo = (uint32_t)(((uint64_t)total_tiles * seg) / p.S);
    const uint32_t t_hi = (uint32_t)(((uint64_t)total_tiles * (seg + 1)) / p.S);
    const size_t ub = (size_t)g * NB * xp.stride + seg;     // this unit's slot of bucket b: ub + b * stride
    // count pass: pairs of the bucket so far; write pass: where the bucket's next run goes, relative to the group's first pair
    if constexpr (WRITE) {
        const uint32_t* st = xp.start + (size_t)blockIdx.x * NB;       // (blockIdx.x = g * S + seg)
        for (uint32_t b = threadIdx.x; b < NB; b += NW * 64) cur[b] = st[b];
    } else {
        for (uint32_t b = threadIdx.x; b < NB; b += NW * 64) cur[b] = 0;
    }                                                                   // (the first tile's barriers order this before any use)
    const uint2* group_pool = WRITE ? p.pool + xp.off[(size_t)g * NB * xp.stride] : nullptr;

    uint2* scratch = scratch_all + (size_t)wave * WS;
    unsigned long long emitted = 0;                         // per lane

    for (uint32_t t = t_lo; t < t_hi; ++t) {
        const uint32_t q = t / p.tiles_per_mat, tile = t - q * p.tiles_per_mat;
        const uint32_t mat = p.gm_list[m0 + q];
        const uint32_t t0 = tile * TW;
        const uint32_t nw = min((uint32_t)TW, p.nwin - t0);
        const uint32_t ncol = nw + K - 1;
        __syncthreads();                                   // previous tile fully consumed
        {
            const float4* src = reinterpret_cast<const float4*>(p.logp + ((size_t)mat * p.sites + t0) * SIGMA);
            float4* dst = reinterpret_cast<float4*>(cols);
            const uint32_t n4 = ncol * (SIGMA / 4);
            for (uint32_t i = threadIdx.x; i < n4; i += NW * 64) dst[i] = src[i];
            const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + t0;
            for (uint32_t i = threadIdx.x; i <= ncol; i += NW * 64) best[i] = bsrc[i];
        }
        __syncthreads();
        for (uint32_t w = wave; w < nw; w += NW) {
            WinCtx c{cols, best, w};
            const uint2 *L, *R;
            uint32_t nL, nR;
            bool ok;
            if constexpr (HalvesDD<SIGMA, K>::OK) ok = build_halves_dd<SIGMA, K, CAP>(c, p.eps, scratch, L, nL, R, nR);
            else ok = build_halves<SIGMA, K, CAP>(c, p.eps, scratch, L, nL, R, nR);
            if (!ok) {                                     // big-list window: queued once, by the count pass
                if (!WRITE && lane == 0) {
                    const uint32_t qi = atomicAdd(p.ovf_count, 1u);
                    p.ovf_queue[qi] = ((unsigned long long)mat << 32) | (unsigned long long)(t0 + w);
                }
                continue;
            }
            if (nL == 0 || nR == 0) continue;
            if (p.flags & 4u) continue;                                        // diagnostics: list building only (nothing is counted or written)
            nL = (uint32_t)__builtin_amdgcn_readfirstlane((int)nL);            // wave-uniform by construction: keep them scalar
            nR = (uint32_t)__builtin_amdgcn_readfirstlane((int)nR);
            // R sorted by score, descending (rank by counting, in place; ties by position).  fl(a + b) is monotone
            // in b, so the pairs of a row that pass `a.score + b.score > eps` (pk_compute.cpp:90-91) are exactly a
            // PREFIX of the sorted R -- the reference's own loop shape (sort at :61-70, break at the first failure,
            // :73-110): a row's count is a binary search, and its pairs leave as one dense run.
            // The count pass needs only the prefix LENGTHS, which do not depend on the order of R: it counts the
            // passing pairs of a row over the unsorted list and skips the sort.
            constexpr int RC = (Geo<SIGMA, K - K / 2, CAP>::CAPH + 63) / 64;
            uint2* Rs = const_cast<uint2*>(R);
            uint32_t rx[RC];
            float ry[RC];
            if constexpr (WRITE) {
                constexpr int RCP = RC <= 1 ? 1 : RC <= 2 ? 2 : RC <= 4 ? 4 : RC <= 8 ? 8 : 16;      // registers of the sorting network: a power of two
                static_assert(RC <= 16, "half lists of at most 1024 entries on the fast path");
                unsigned long long kj[RCP];
#pragma unroll
                for (int ch = 0; ch < RCP; ++ch) {
                    const uint32_t j = (uint32_t)ch * 64 + lane;
                    kj[ch] = 0;
                    if (ch < RC && j < nR) kj[ch] = ((unsigned long long)enc_score_bits(R[j].y) << 32) | (unsigned long long)(0xFFFFFFFFu - j);
                }
                // (score, position) keys sorted in registers (bitonic network over as many 64-key registers as the list needs)
                const uint32_t nch = (nR + 63) >> 6;
                if (nch <= 1) xp_sort_desc<1, RCP>(kj);
                else if (nch <= 2) { if constexpr (RCP >= 2) xp_sort_desc<2, RCP>(kj); }
                else if (nch <= 4) { if constexpr (RCP >= 4) xp_sort_desc<4, RCP>(kj); }
                else if (nch <= 8) { if constexpr (RCP >= 8) xp_sort_desc<8, RCP>(kj); }
                else { if constexpr (RCP >= 16) xp_sort_desc<16, RCP>(kj); }
                // the entry of rank e comes from position j = ~key.low of the unsorted list: gathered, then written back in order
                uint32_t ryb[RC];
#pragma unroll
                for (int ch = 0; ch < RC; ++ch) {
                    const uint32_t e = (uint32_t)ch * 64 + lane;
                    rx[ch] = 0; ryb[ch] = 0; ry[ch] = 0.f;
                    if (e < nR) { const uint2 b = R[0xFFFFFFFFu - (uint32_t)kj[ch]]; rx[ch] = b.x; ryb[ch] = b.y; ry[ch] = __uint_as_float(b.y); }
                }
                wave_lds_sync();                           // every lane has read the unsorted list
#pragma unroll
                for (int ch = 0; ch < RC; ++ch)
                    if ((uint32_t)ch * 64 + lane < nR) Rs[(uint32_t)ch * 64 + lane] = make_uint2(rx[ch], ryb[ch]);
                wave_lds_sync();
            }
            const uint32_t P2 = 1u << (31 - __builtin_clz(nR));                // largest power of two <= nR
            for (uint32_t ib = 0; ib < nL; ib += 64) {
                // one row per lane: the length of the passing prefix, by binary lifting
                const bool vr = ib + lane < nL;
                uint2 a = make_uint2(0, 0);
                if (vr) a = L[ib + lane];
                const float ay = __uint_as_float(a.y);
                uint32_t cnt = 0;
                if constexpr (WRITE) {
                    for (uint32_t st = P2; st > 0; st >>= 1) {
                        const uint32_t probe = cnt + st;
                        if (probe <= nR) {
                            const float s = ay + __uint_as_float(Rs[probe - 1].y);  // pk_compute.cpp:90
                            if (s > p.eps) cnt = probe;                             // :91
                        }
                    }
                } else {
                    // (sixteen columns' broadcast reads in flight per LDS round trip: with four, a 135-entry list cost 34 exposed
                    //  round trips per row block)
                    uint32_t j = 0;
                    for (; j + 16 <= nR; j += 16) {
                        float by[16];
#pragma unroll
                        for (uint32_t u = 0; u < 16; ++u) by[u] = __uint_as_float(R[j + u].y);
#pragma unroll
                        for (uint32_t u = 0; u < 16; ++u) cnt += (ay + by[u] > p.eps) ? 1u : 0u;      // pk_compute.cpp:90-91
                    }
#pragma unroll 4
                    for (; j < nR; ++j) {
                        const float s = ay + __uint_as_float(R[j].y);              // pk_compute.cpp:90
                        cnt += (s > p.eps) ? 1u : 0u;                              // :91
                    }
                }
                if (!vr) cnt = 0;
                const uint32_t bk = a.x / RPB;
                if constexpr (!WRITE) {
                    emitted += cnt;
                    if (cnt) atomicAdd(&cur[bk], cnt);
                } else {
                    // the bucket's cursor started at the unit's scan offset; every row's run as a byte address
                    unsigned long long run = 0;
                    if (cnt) run = reinterpret_cast<unsigned long long>(group_pool + atomicAdd(&cur[bk], cnt));
                    const uint32_t axm = a.x * mulR;
                    const uint32_t rows = to_sgpr(min(64u, nL - ib));
                    // One row at a time, its values broadcast with v_readlane; the row's passing prefix leaves through the
                    // scalar-base store form with exec written directly (the loop is ~30 instructions per row otherwise, most of
                    // them scalar address arithmetic and exec bookkeeping -- and this loop is half of the kernel's instructions)
                    auto do_row = [&](uint32_t r) {
                        const uint32_t cr = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)r);
                        const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)run, (int)r);
                        const uint32_t rhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(run >> 32), (int)r);
                        const uint2* dst = reinterpret_cast<const uint2*>(((unsigned long long)rhi << 32) | rlo);
                        const uint32_t ax = (uint32_t)__builtin_amdgcn_readlane((int)axm, (int)r);
                        const float ayr = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)a.y, (int)r));
                        auto chunk = [&](auto CHK) {
                            constexpr int ch = decltype(CHK)::value;
                            if constexpr (ch < RC) {
                                const uint32_t left = cr - (uint32_t)ch * 64u;                       // (> 0 here)
                                const uint64_t mask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
                                store8_lanes<ch * 512>(dst, lane8, ax + rx[ch], __float_as_uint(ayr + ry[ch]), mask);   // :90, the row's passing prefix
                            }
                        };
                        if (cr < 64) {                                  // the common case: one store, its lanes = the low cr bits
                            unsigned long long mask;                    // (cr == 0: no lanes, the store is a no-op)
                            asm("s_bfm_b64 %0, %1, 0" : "=s"(mask) : "s"(cr));
                            store8_lanes<0>(dst, lane8, ax + rx[0], __float_as_uint(ayr + ry[0]), mask);
                            return;
                        }
                        chunk(std::integral_constant<int, 0>{});
                        if (cr > 64) {
                            chunk(std::integral_constant<int, 1>{});
                            if (cr > 128) {
                                chunk(std::integral_constant<int, 2>{});
                                if (cr > 192) {
                                    chunk(std::integral_constant<int, 3>{});
                                    if (cr > 256) {
                                        chunk(std::integral_constant<int, 4>{});
                                        if (cr > 320) chunk(std::integral_constant<int, 5>{});
                                        if (cr > 384) chunk(std::integral_constant<int, 6>{});
                                        if (cr > 448) chunk(std::integral_constant<int, 7>{});
                                    }
                                }
                            }
                        }
                    };
                    // (two rows per trip measured slower: 20.2 against 19.4 ms at cfg4)
                    for (uint32_t r = 0; r < rows; ++r) do_row(r);
                }
            }
        }
    }
    if constexpr (!WRITE) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < NB; b += NW * 64) xp.cnt[ub + (size_t)b * xp.stride] = cur[b];
        // wave sum of the per-lane counts
        for (int o = 32; o > 0; o >>= 1) emitted += __shfl_down(emitted, o, 64);
        if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);
    }
}

// Big-list windows of the exact-partition variant (queued by the count pass of score_xp_kernel): same cooperative
// shape as score_overflow_kernel, run twice like the fast path.  Their pairs take segment S of every (group, bucket):
// the count pass adds a row's passing pairs to that slot (one global atomic per row), the write pass reserves the
// row's run behind the slot's offset and writes it.  So these pairs, too, are max-reduced in LDS, and nothing
// touches the tables after the reduce pass.
template <int SIGMA, int K, uint32_t TBL, bool WRITE>
__global__ __launch_bounds__(OVF_NW * 64) void score_overflow_xp_kernel(XpParams xp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const StreamParams& p = xp.s;
    constexpr int CAPF = big_capf<SIGMA, K>();
    constexpr uint32_t T = ipow(SIGMA, K);
    constexpr uint32_t NB = (T + TBL - 1) / TBL;
    constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
    static_assert(TBL % mulR == 0, "a row of the final join must stay inside one bucket");
    constexpr uint32_t RPB = TBL / mulR;
    using TG = TileGeo<SIGMA, K, 1>;
    __shared__ uint32_t sh_n[2];
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + TG::COLS_F;
    uint2* scratch = reinterpret_cast<uint2*>(smem + TG::HEAD_BYTES);
    const uint32_t n = *p.ovf_count;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long emitted = 0;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const unsigned long long e = p.ovf_queue[q];
        const uint32_t mat = (uint32_t)(e >> 32), start = (uint32_t)e;
        const uint32_t g = p.mat_slot[mat];
        __syncthreads();                                                       // previous window's lists consumed
        const float* src = p.logp + ((size_t)mat * p.sites + start) * SIGMA;
        for (uint32_t i = threadIdx.x; i < K * SIGMA; i += OVF_NW * 64) cols[i] = src[i];
        const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + start;
        for (uint32_t i = threadIdx.x; i <= K; i += OVF_NW * 64) best[i] = bsrc[i];
        __syncthreads();
        const uint2 *L = scratch, *R = scratch + Geo<SIGMA, K / 2, CAPF>::CAPH;
        if (wave == 0) {
            WinCtx c{cols, best, 0};
            uint32_t nL = 0, nR = 0;
            const bool fits = build_halves<SIGMA, K, CAPF>(c, p.eps, scratch, L, nL, R, nR);      // (cannot overflow at full capacity)
            if (!fits) { nL = 0; nR = 0; if (lane == 0 && p.big_ovf) atomicOr(p.big_ovf, 1u); }   // capped lists (big_capf): the call fails
            if (lane == 0) { sh_n[0] = nL; sh_n[1] = nR; }
        }
        __syncthreads();
        const uint32_t nL = sh_n[0], nR = sh_n[1];
        if (nL == 0 || nR == 0) continue;
        // Rows of L in batches of XP_ROWS per wavefront: the batch's counts over one walk of R (a block of R in registers, the rows'
        // values read once), then ONE global atomic instruction for the batch -- lane u reserves row u's run -- where every row used to
        // wait for a returning atomic of its own (and for its offset's load), then the rows' stores.
        constexpr uint32_t XP_ROWS = 8;
        for (uint32_t i0 = wave * XP_ROWS; i0 < nL; i0 += OVF_NW * XP_ROWS) {
            const uint32_t nrow = min(XP_ROWS, nL - i0);
            uint2 a[XP_ROWS];
            uint32_t c[XP_ROWS];
#pragma unroll
            for (uint32_t u = 0; u < XP_ROWS; ++u) { a[u] = L[min(i0 + u, nL - 1)]; c[u] = 0; }
            for (uint32_t jb = 0; jb < nR; jb += 64) {
                const uint32_t j = jb + lane;
                const bool valid = j < nR;
                const float by = __uint_as_float(valid ? R[j].y : 0u);
#pragma unroll
                for (uint32_t u = 0; u < XP_ROWS; ++u) {
                    const float s = __uint_as_float(a[u].y) + by;                                  // pk_compute.cpp:90
                    c[u] += (uint32_t)__popcll(__ballot(valid && s > p.eps));                      // :91
                }
            }
            // lane u holds row u's count and slot
            uint32_t my_c = 0, my_ax = 0;
#pragma unroll
            for (uint32_t u = 0; u < XP_ROWS; ++u)
                if (lane == u) { my_c = u < nrow ? c[u] : 0u; my_ax = a[u].x; }
            const uint32_t my_bk = my_ax / RPB;
            const size_t my_slot = ((size_t)g * NB + my_bk) * xp.stride + (xp.stride - 1);
            if constexpr (!WRITE) {
                if (my_c) atomicAdd(&xp.cnt[my_slot], my_c);
                emitted += my_c;                                                                   // (per lane; summed at the end)
            } else {
                unsigned long long my_dst = 0;
                if (my_c) my_dst = reinterpret_cast<unsigned long long>(p.pool + xp.off[my_slot] + atomicAdd(&xp.ovcur[(size_t)g * NB + my_bk], my_c));
#pragma unroll 1
                for (uint32_t u = 0; u < nrow; ++u) {
                    const uint32_t cu = (uint32_t)__builtin_amdgcn_readlane((int)my_c, (int)u);
                    if (cu == 0) continue;
                    const uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_dst, (int)u);
                    const uint32_t dhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_dst >> 32), (int)u);
                    // (a pointer rebuilt from lane values: told to be global memory, or the stores come out as flat_store -- tests/test_isa.py)
                    typedef unsigned long long __attribute__((address_space(1)))* global_pair_ptr;
                    const global_pair_ptr dst = (global_pair_ptr)(((unsigned long long)dhi << 32) | dlo);
                    const uint2 au = L[i0 + u];
                    uint32_t done = 0;
                    for (uint32_t jb = 0; jb < nR; jb += 64) {
                        const uint32_t j = jb + lane;
                        uint2 b = make_uint2(0, 0);
                        if (j < nR) b = R[j];
                        const float s = __uint_as_float(au.y) + __uint_as_float(b.y);
                        const bool pass = j < nR && s > p.eps;
                        const uint64_t m = __ballot(pass);
                        if (pass) dst[done + mbcnt(m)] = (unsigned long long)(au.x * mulR + b.x) | ((unsigned long long)__float_as_uint(s) << 32);
                        done += (uint32_t)__popcll(m);
                    }
                }
            }
        }
    }
    if constexpr (!WRITE) {
        for (int o = 32; o > 0; o >>= 1) emitted += __shfl_down(emitted, o, 64);                  // (per-lane counts: lanes 0..XP_ROWS-1)
        if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);
    }
}

// xp pass 3: one workgroup per (group, bucket); its pairs are ONE contiguous range of the pool
// (segments of a (group, bucket) are adjacent in the scan order).
//
// COMPRESS: for sparse tables (AA k=6: 28 % of 64 M slots per group) the dense slice is NOT written.  The group's
// table is kept as  occupancy bits (mask)  +  the non-empty slots' score codes in slot order, written in place over
// the head of the (group, bucket) pool range just consumed (unique slots <= pairs)  +  per 64-slot block the number
// of non-empty slots before it in the slice (rank).  Slot x of group g then sits at
//   values(g, b)[rank[g][x / 64] + popcount(mask64[g][x / 64] & below(x % 64))],  b = x / TBL   (vaddr[g][x / 64] = &values(g, b)[rank[g][x / 64]]),
// values(g, b) = (u32*)(pool + off[(g * NB + b) * S]).  Consumers: km_write_c_kernel, write_chunks_c_kernel.
template <uint32_t TBL, int NT, bool COMPRESS>
__global__ __launch_bounds__(NT) void reduce_ranges_kernel(uint2* __restrict__ pool, const uint64_t* __restrict__ off,
                                                          uint32_t S, uint32_t NB, uint64_t T, uint32_t* __restrict__ table,
                                                          uint32_t* __restrict__ mask, uint64_t mask_words,
                                                          uint32_t* __restrict__ rank, uint64_t* __restrict__ vaddr,
                                                          uint32_t* __restrict__ ucnt)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
    const uint32_t gb = blockIdx.x;
    const uint32_t g = gb / NB, b = gb - g * NB;
    const uint64_t key0 = (uint64_t)b * TBL;
    const uint32_t nslots = (uint32_t)min((uint64_t)TBL, T - key0);
    const uint64_t r0 = off[(size_t)gb * S], r1 = off[(size_t)gb * S + S];     // S = slots per (group, bucket) in the scan
    constexpr int PER = 8;
    uint2 v[PER];
    uint64_t i = r0 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < PER; ++j) if (i + (uint64_t)j * NT < r1) v[j] = pool_load(pool + i + (uint64_t)j * NT);   // in flight while the table is cleared
    constexpr uint32_t CLR = COMPRESS ? comp_padded_slots<TBL, NT>() : TBL;      // (the compress epilogue reads whole blocks per wavefront)
    for (uint32_t z = threadIdx.x; z < CLR / 4; z += NT) reinterpret_cast<uint4*>(tab)[z] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t k0 = (uint32_t)key0;
    while (i < r1) {
        uint2 cur[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) cur[j] = v[j];
        const uint64_t ci = i;
        i += (uint64_t)PER * NT;
#pragma unroll
        for (int j = 0; j < PER; ++j) if (i + (uint64_t)j * NT < r1) v[j] = pool_load(pool + i + (uint64_t)j * NT);
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (ci + (uint64_t)j * NT < r1) atomicMax(&tab[cur[j].x - k0], enc_score_bits(cur[j].y));
    }
    __syncthreads();
    if constexpr (!COMPRESS) {
        uint32_t* dst = table + (size_t)g * T + key0;
        if ((nslots & 3u) == 0 && ((((size_t)g * T + key0) & 3u) == 0)) {
            for (uint32_t z = threadIdx.x; z < nslots / 4; z += NT) reinterpret_cast<uint4*>(dst)[z] = reinterpret_cast<uint4*>(tab)[z];
        } else {
            for (uint32_t z = threadIdx.x; z < nslots; z += NT) dst[z] = tab[z];
        }
        if (mask) store_slice_mask(tab, nslots, mask + (size_t)g * mask_words + (key0 >> 5), NT);
    } else {
        // in place: every pair of the range has been consumed
        compress_slice<TBL, NT>(tab, nslots, reinterpret_cast<uint32_t*>(pool + r0), mask + (size_t)g * mask_words + (key0 >> 5),
                                rank + (size_t)g * (mask_words / 2) + (key0 >> 6), vaddr + (size_t)g * (mask_words / 2) + (key0 >> 6), ucnt + gb);
    }
}

}  // namespace ipkgpu
