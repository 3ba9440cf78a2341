// kernels_reduce_pipe.hpp -- pass 2 of the stream variant for 128-KB table slices (DNA k = 11, 12: TBL = 32768 slots), persistent.
//
// ipk::put (ipk/src/branch_group.cpp:88-101) for the pairs of one (group, key bucket): LDS max-reduce of the bucket's chunks, then
// the slice's compressed form -- what reduce_buckets_kernel<TBL, NT, true> computes, bit for bit.  That kernel is one workgroup per
// slice; with a 128-KB table only ONE workgroup fits a CU, so its phases -- chunk descriptors -> pair loads -> LDS atomics ->
// compress -> stores -- ran strictly one after the other on every CU: r04_cfg3_share_backhalf_sq.json shows 0.10 vector-memory
// instructions in flight per wavefront cycle, 47 % of wavefront time in waits, and a slice whose ~130 chunks (most of them the
// half-filled last chunks of the scoring wavefronts) went through four dependent load round trips: 14.8 us per slice where the
// HBM share of a CU would allow ~9.5.
//
// Here a workgroup stays on its CU and walks slices blockIdx.x, + gridDim.x, ...; per wavefront the pair loads run as a stream of
// TRIPS (up to D chunks, D * CH / 64 loads per lane) through two register buffers: while trip n is reduced into the table, trip
// n + 1 is in flight, and trip n + 2 is issued as soon as n's registers are free -- across slice boundaries, so the compress
// epilogue of slice i runs under the loads of slice i + 1.  A wavefront's share of a slice is a contiguous range of the slice's
// chunk list; the D descriptors of a trip come in by ONE scalar load requested a trip ahead, the slice offsets by scalar loads
// two slices ahead -- nothing but the pair loads (and the epilogue's stores) counts on vmcnt.  Every trip issues the same number of
// loads (absent chunks and the empty part of a half-filled chunk re-read the chunk's first line), so the waits are counted
// vmcnt(N), never 0.
#pragma once
#include "kernels_score.hpp"
#include "kernels_keymajor.hpp"

namespace ipkgpu {

template <uint32_t TBL, int NT, int D>
__global__ __launch_bounds__(NT) void reduce_buckets_pipe_kernel(const uint2* __restrict__ pool, const uint64_t* __restrict__ off,
                                                                 const uint2* __restrict__ list, uint32_t n_gb, uint32_t NB, uint64_t T,
                                                                 uint32_t* __restrict__ mask, uint64_t mask_words,
                                                                 uint2* __restrict__ cvals, const uint64_t* __restrict__ coff,
                                                                 uint32_t* __restrict__ rank, uint64_t* __restrict__ vaddr,
                                                                 uint32_t* __restrict__ ucnt)
{
    constexpr uint32_t CH = chunk_pairs<TBL>();
    constexpr int PER = CH / 64;                            // loads per lane and chunk
    constexpr uint32_t NWV = NT / 64, PAD = comp_padded_slots<TBL, NT>();
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(smem);
    const uint32_t lane = lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t G = gridDim.x, first = blockIdx.x;
    if (first >= n_gb) return;
    const uint32_t n_items = (n_gb - first + G - 1) / G;    // slices of this workgroup: first, first + G, ...

    // ---- slice metadata (wave-uniform: scalar loads) ------------------------------------------------------------------------
    struct Meta { uint32_t c0, n; };                        // first chunk-list entry, chunks
    auto load_meta = [&](uint32_t x) {
        Meta m{0u, 0u};
        if (x < n_items) {
            const uint32_t gb = first + x * G;
            const uint64_t o0 = uniform_load(off + gb), o1 = uniform_load(off + gb + 1);
            m.c0 = (uint32_t)o0; m.n = (uint32_t)(o1 - o0);
        }
        return m;
    };
    auto load_cv = [&](uint32_t x) { return x < n_items ? uniform_load(coff + (first + x * G)) : 0ull; };   // offset of the slice's values
    // this wavefront's share of a slice's chunks: a contiguous range, the shares differ by at most one chunk
    auto share = [&](const Meta& m, uint32_t& start, uint32_t& cnt) {
        const uint32_t q = m.n / NWV, r = m.n % NWV;
        cnt = q + (wave < r ? 1u : 0u);
        start = m.c0 + wave * q + min(wave, r);
    };
    // the descriptors (chunk id, pairs) of the D chunks of a trip: one scalar load (the list is padded by D entries)
    struct Desc { uint2 e[D]; };
    static_assert(D == 2 || D == 4 || D == 8, "a trip's descriptors are one s_load_dwordx{4,8,16}");
    auto load_desc = [&](uint32_t idx) {                     // (adjacent scalar loads: merged into one)
        Desc d;
        const uint64_t* p = reinterpret_cast<const uint64_t*>(list + idx);
#pragma unroll
        for (int j = 0; j < D; ++j) { const uint64_t w = uniform_load(p + j); d.e[j] = make_uint2((uint32_t)w, (uint32_t)(w >> 32)); }
        return d;
    };

    uint32_t cur = 0, lpos = 0, cstart, ccnt;                // load cursor: slice cur, chunk lpos of this wavefront's share [cstart, + ccnt)
    Meta mcur = load_meta(0), mnext = load_meta(1), min2 = load_meta(2);
    share(mcur, cstart, ccnt);
    // the NEXT trip: its descriptors are requested one trip ahead (kind: 0 = nothing, the cursor is past the last slice; 1 = chunks
    // of a slice; 2 = ... and the share's last ones, possibly none)
    uint32_t ntake = min((uint32_t)D, ccnt), nkind = ccnt <= (uint32_t)D ? 2u : 1u;
    Desc nd = load_desc(cstart);
    uint64_t cv = load_cv(0), cvn = load_cv(1);              // offsets of this slice's values, of the next one's

    uint32_t it = 0;                                         // the slice being reduced
    uint32_t k0 = 0;

    // ---- trips ----------------------------------------------------------------------------------------------------------------
    auto fill = [&](uint2 (&P)[D * PER], uint32_t (&pc)[D]) -> uint32_t {
        const uint32_t kind = nkind, take = ntake;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const bool ok = (uint32_t)j < take;
            const uint32_t id = ok ? nd.e[j].x : 0u, cnt = ok ? nd.e[j].y : 0u;
            pc[j] = cnt;
            const uint2* s = pool + (size_t)id * CH;
#pragma unroll
            for (int q = 0; q < PER; ++q) {                  // (the select is wave-uniform: a scalar base per load, one lane offset for all)
                const uint2* sq = s + ((uint32_t)(64 * q) < cnt ? 64u * q : 0u);
                P[j * PER + q] = pool_load(sq + lane);
            }
        }
        // the cursor moves on; the next trip's descriptors are requested
        lpos += take;
        if (kind == 2) {
            ++cur; lpos = 0;
            mcur = mnext; mnext = min2; min2 = load_meta(cur + 2);
            share(mcur, cstart, ccnt);
        }
        if (kind != 0 && cur < n_items) {
            ntake = min((uint32_t)D, ccnt - lpos);
            nkind = lpos + ntake >= ccnt ? 2u : 1u;
            nd = load_desc(cstart + lpos);
        } else {
            ntake = 0; nkind = 0;
        }
        return kind;
    };
    auto consume = [&](const uint2 (&P)[D * PER], const uint32_t (&pc)[D]) {
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
            for (int q = 0; q < PER; ++q)
                if (lane + 64u * q < pc[j]) atomicMax(&tab[P[j * PER + q].x - k0], enc_score_bits(P[j * PER + q].y));
    };
    // the end of slice `it` for this wavefront: every wavefront comes here once per slice (three barriers each time)
    auto boundary = [&]() {
        __syncthreads();                                     // the slice's atomics have landed
        const uint32_t gb = first + it * G;
        const uint32_t g = gb / NB, b = gb - g * NB;
        const uint64_t key0 = (uint64_t)b * TBL;
        const uint32_t nslots = (uint32_t)min((uint64_t)TBL, T - key0);
        compress_slice<TBL, NT, true, true>(tab, nslots, reinterpret_cast<uint32_t*>(cvals + cv), mask + (size_t)g * mask_words + (key0 >> 5),
                                            rank + (size_t)g * (mask_words / 2) + (key0 >> 6), vaddr + (size_t)g * (mask_words / 2) + (key0 >> 6), ucnt + gb);
        __syncthreads();                                     // the table is clean again
        ++it;
        cv = cvn; cvn = load_cv(it + 1);
        const uint32_t gbn = first + it * G;
        k0 = (gbn % NB) * TBL;
    };

    k0 = (first % NB) * TBL;
    uint2 PA[D * PER], PB[D * PER];
    uint32_t pcA[D], pcB[D];
    uint32_t kA = fill(PA, pcA), kB = fill(PB, pcB);
    for (uint32_t i = threadIdx.x; i < PAD / 4; i += NT) reinterpret_cast<uint4*>(tab)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // (the loop is rotated -- its head is a fill, not a consume: the compiler's wait-count pass is exact inside straight-line code but
    //  falls back to "nothing newer is in flight" for registers consumed right at a loop head, which would wait for BOTH trips)
    consume(PA, pcA);
    bool endA = kA == 2;
    for (;;) {
        kA = fill(PA, pcA);
        if (endA) { boundary(); if (it >= n_items) break; }
        consume(PB, pcB);
        const bool endB = kB == 2;
        kB = fill(PB, pcB);
        if (endB) { boundary(); if (it >= n_items) break; }
        consume(PA, pcA);
        endA = kA == 2;
    }
}

}  // namespace ipkgpu
