// kernels_quad.hpp -- pass 1 of the stream variant for DNA k = 8..12, rebuilt around VALU instruction count.
//
// score_stream_kernel (kernels_score.hpp) is bound by VALU issue: one vector instruction per SIMD every four
// cycles, ~440 of them per window at cfg2, most of them with a quarter of the lanes doing useful work.  This
// kernel computes the same sets with the same float operations (pk_compute.cpp:42-114) and appends them to the same
// pair pool, but spends about half the instructions per window:
//
//   thresholds   every hierarchical bound of a window -- eps - (best[a] - best[b]) chains, pk_compute.cpp:54-55 --
//                is window-uniform: one thread per window computes the window's 22 bounds once per tile into LDS;
//                the scoring lanes fetch them with broadcast reads (LDS instructions, not VALU).
//   child nodes  the 2- and 3-symbol nodes are evaluated for FOUR windows per wavefront, 16 lanes each: a
//                2-symbol node is one step for four windows; a 3-symbol node evaluates its inner pair once and
//                loops over the four states of its first column.  Survivors are compacted per window with a
//                16-bit field of the ballot.
//   half joins   one window at a time, both halves in one 64-candidate step (as before).
//   final join   R (<= 32 entries in the common case) is held in registers, floor(64 / |R|) rows of L per step:
//                no index arithmetic per step, and every lane group works on ONE row, hence one key bucket.
//   append       slots are reserved per ROW: the row's first lane adds the row's survivor count to the bucket's
//                {fill, chunk} word pair with one 64-bit LDS atomic and the result is broadcast to the row's
//                lanes (ds_bpermute) -- a handful of atomics per step instead of one per pair.
//
// Windows whose lists exceed the fast path's capacity are queued for the big-list kernel exactly as before.
#pragma once
#include "kernels_score.hpp"

namespace ipkgpu {

// The four child nodes of a window: LA = (0, LA), LB = (LA, LB), RA = (HL, RA), RB = (HL + RA, RB), each of 2 or 3 symbols.
// A wavefront evaluates them for TWO windows at a time in 16-lane slots: slot = (window half, node of the step), so a
// step runs two nodes of equal size side by side (k = 10: step 0 = LA | RA, step 1 = LB | RB).
struct QuadNode { int J, H, TH, OFF; uint32_t F, CMUL; };
struct QuadStep { int H, A, B; };                                  // nodes of the step (B = -1: the second slot idles)

template <int SIGMA, int K>
struct QuadGeo {
    using D = HalvesDD<SIGMA, K>;
    static constexpr int HL = D::HL, HR = D::HR, LA = D::LA, LB = D::LB, RA = D::RA, RB = D::RB;
    static constexpr bool OK = SIGMA == 4 && D::OK && LA >= 2 && LA <= 3 && LB >= 2 && LB <= 3 && RA >= 2 && RA <= 3 && RB >= 2 && RB <= 3;
    static constexpr uint32_t FLA = D::FLA, FLB = D::FLB, FRA = D::FRA, FRB = D::FRB;
    static constexpr uint32_t mulR = ipow(SIGMA, K - K / 2);
    static constexpr uint32_t CW = FLA + FLB + FRA + FRB;          // child entries per window
    static constexpr int th_n(int h) { return h == 2 ? 3 : 5; }    // thresholds of a node: itself, its leaves (and inner pair)
    static constexpr int TH0 = 2, TH1 = TH0 + th_n(LA), TH2 = TH1 + th_n(LB), TH3 = TH2 + th_n(RA);
    static constexpr int TH_F = TH3 + th_n(RB);                    // floats per window in the threshold table (eps_l, eps_r, 4 nodes)
    static constexpr int PADW = 1;                                 // a pair may reach one window past the tile's last one
    static constexpr QuadNode node(int n)
    {
        // LB's codes carry mulR, so that L's codes come out multiplied by it (final code = L.code + R.code)
        return n == 0 ? QuadNode{0, LA, TH0, 0, FLA, 1u}
             : n == 1 ? QuadNode{LA, LB, TH1, (int)FLA, FLB, mulR}
             : n == 2 ? QuadNode{HL, RA, TH2, (int)(FLA + FLB), FRA, 1u}
                      : QuadNode{HL + RA, RB, TH3, (int)(FLA + FLB + FRA), FRB, 1u};
    }
    // steps: the 2-symbol nodes paired in node order, then the 3-symbol nodes
    static constexpr int count_h(int h) { int c = 0; for (int n = 0; n < 4; ++n) c += node(n).H == h; return c; }
    static constexpr int nth_h(int h, int i) { for (int n = 0; n < 4; ++n) if (node(n).H == h) { if (i == 0) return n; --i; } return -1; }
    static constexpr int NS2 = (count_h(2) + 1) / 2, NS3 = (count_h(3) + 1) / 2, NSTEPS = NS2 + NS3;
    static constexpr QuadStep step(int s)
    {
        return s < NS2 ? QuadStep{2, nth_h(2, 2 * s), nth_h(2, 2 * s + 1)} : QuadStep{3, nth_h(3, 2 * (s - NS2)), nth_h(3, 2 * (s - NS2) + 1)};
    }
    static constexpr int step_of(int n) { for (int s = 0; s < NSTEPS; ++s) if (step(s).A == n || step(s).B == n) return s; return -1; }
    static constexpr int slot_of(int n) { return step(step_of(n)).B == n ? 1 : 0; }
    // ONE window per wavefront step (k = 11, 12: half the child-list scratch): the four 16-lane slots hold up to four nodes of equal
    // size of the same window -- the 2-symbol nodes in one step, the 3-symbol nodes in another
    static constexpr int NS1 = (count_h(2) > 0 ? 1 : 0) + (count_h(3) > 0 ? 1 : 0);
    static constexpr int step1_h(int s) { return (s == 0 && count_h(2) > 0) ? 2 : 3; }
    static constexpr int step1_node(int s, int slot) { return nth_h(step1_h(s), slot); }        // -1: the slot idles
    static constexpr int step1_of(int n) { return node(n).H == 2 ? 0 : (count_h(2) > 0 ? 1 : 0); }
    static constexpr int slot1_of(int n) { int c = 0; for (int m = 0; m < n; ++m) c += node(m).H == node(n).H; return c; }
};

template <int SIGMA, int K, int TW>
struct QuadTile {
    static constexpr int TC = TW + K - 1 + QuadGeo<SIGMA, K>::PADW;
    static constexpr int COLS_F = TC * SIGMA;
    static constexpr int BEST_F = ((TC + 1 + 3) / 4) * 4;
    static constexpr int TH_FLOATS = (((TW + QuadGeo<SIGMA, K>::PADW) * QuadGeo<SIGMA, K>::TH_F + 3) / 4) * 4;
    static constexpr int HEAD_BYTES = (COLS_F + BEST_F + TH_FLOATS) * 4;
};

template <int SIGMA, int K, int CAP, bool ONEWIN = false>
constexpr uint32_t quad_wave_entries()
{
    using Q = QuadGeo<SIGMA, K>;
    return (ONEWIN ? 1 : 2) * Q::CW + Geo<SIGMA, Q::HL, CAP>::CAPH + Geo<SIGMA, Q::HR, CAP>::CAPH;
}

// Thresholds of one node (J, H) under the node threshold e (Direct<>::eval's own expressions, dcla_device.hpp).
template <int J, int H>
__device__ __forceinline__ void quad_node_thresholds(const float* bs, float e, float* out)
{
    out[0] = e;
    if constexpr (H == 2) {
        out[1] = e - (bs[J + 2] - bs[J + 1]);                      // left leaf:  eps - M(right)   (:54)
        out[2] = e - (bs[J + 1] - bs[J]);                          // right leaf: eps - M(left)    (:55)
    } else {
        static_assert(H == 3, "child nodes have 2 or 3 symbols");
        const float e_leaf = e - (bs[J + 3] - bs[J + 1]);          // first column (HL = 1)
        const float e_pair = e - (bs[J + 1] - bs[J]);              // columns J+1, J+2
        out[1] = e_leaf;
        out[2] = e_pair;
        out[3] = e_pair - (bs[J + 3] - bs[J + 2]);                 // pair's left leaf
        out[4] = e_pair - (bs[J + 2] - bs[J + 1]);                 // pair's right leaf
    }
}

// One step: the H-symbol node of every 16-lane slot.  `base` = LDS float index of the node's first column of the
// slot's window (cols[(window + J) * 4]); th = the node's thresholds; list = the node's list of the slot's window;
// code16 = l16 * cmul.  Sub-step i's survivors follow sub-step i-1's in the list (ascending code order).
template <int H>
__device__ __forceinline__ void quad_step(const float* cols, uint32_t base, bool valid, uint64_t vmask, const float* th, uint2* list,
                                          uint32_t l16, uint32_t slot, uint32_t lt16, uint32_t code16, uint32_t cmul, uint32_t& count)
{
    // count: survivors of this lane's slot (the node's list length)
    const float e = th[0];
    if constexpr (H == 2) {
        const float sl = cols[base + (l16 >> 2)], sr = cols[base + 4 + (l16 & 3u)];
        const float s = sl + sr;                                                   // pk_compute.cpp:90
        const bool c1 = sl > th[1], c2 = sr > th[2], c3 = s > e;                   // as_column x2, :91
        const bool pass = valid & c1 & c2 & c3;
        const uint64_t m = vmask & ballot64(c1) & ballot64(c2) & ballot64(c3);     // (ballots of the single compares: no VALU)
        const uint32_t x = (uint32_t)(m >> (16u * slot));
        const uint32_t pos = (uint32_t)__popc(x & lt16);
        if (pass) list[pos] = make_uint2(code16, __float_as_uint(s));
        count = (uint32_t)__popc(x & 0xFFFFu);
    } else {
        const float sj = cols[base + 4 + (l16 >> 2)], sl2 = cols[base + 8 + (l16 & 3u)];
        const float s2 = sj + sl2;                                                 // the inner pair (J+1, J+2)
        const bool d1 = sj > th[3], d2 = sl2 > th[4], d3 = s2 > th[2];
        const bool okp = valid & d1 & d2 & d3;
        const uint64_t mp = vmask & ballot64(d1) & ballot64(d2) & ballot64(d3);
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float si = cols[base + i];
            const float s = si + s2;                                               // :90
            const bool c1 = si > th[1], c2 = s > e;                                // :91
            const bool pass = okp & c1 & c2;
            const uint64_t m = mp & ballot64(c1) & ballot64(c2);
            const uint32_t x = (uint32_t)(m >> (16u * slot));
            const uint32_t pos = (uint32_t)__popc(x & lt16) + run;
            if (pass) list[pos] = make_uint2(__umul24(cmul, (uint32_t)i * 16u) + code16, __float_as_uint(s));
            run += (uint32_t)__popc(x & 0xFFFFu);
        }
        count = run;
    }
}

// A pool store: wave-uniform base (SGPR pair) + 32-bit byte offset, i.e. the `global_store v_off, v_data, s[base]` form -- one
// VALU instruction of address arithmetic per store instead of four for a 64-bit lane address.  The offsets are relative to the
// wave's base chunk (RowAppender::rebase keeps every open chunk within 4 GiB of it).  (A structured-buffer form -- buffer_store
// ... idxen offen with one record per chunk -- was tried and is WRONG here: the buffer unit forms index * stride + offset in 32
// bits relative to the POOL's base, so chunks past 4 GiB wrap around.  The pool is tens of GB.)
__device__ __forceinline__ void pool_store(uint2* wave_base, uint32_t byte_off, uint2 val)
{
    const unsigned long long data = ((unsigned long long)val.y << 32) | val.x;
    asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(byte_off), "v"(data), "s"(wave_base) : "memory");
}
// (x << 3) + y as ONE instruction (left to itself the compiler shares x << 3 with the rarely taken chunk-roll path and adds separately)
__device__ __forceinline__ uint32_t lshl3_add(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// The same store for the lanes of `mask`, from code that runs with ALL lanes enabled (uniform control flow in full wavefronts):
// two scalar instructions around the store instead of the compiler's and-saveexec / branch-if-empty / restore sequence --
// the kernel is as short of scalar issue slots as of vector ones.
__device__ __forceinline__ void pool_store_lanes(uint2* wave_base, uint32_t byte_off, uint2 val, uint64_t mask)
{
    const unsigned long long data = ((unsigned long long)val.y << 32) | val.x;
    IPK_ASSERT_FULL_EXEC();
    asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_mov_b64 exec, -1"
                 : : "v"(byte_off), "v"(data), "s"(wave_base), "s"(mask) : "memory");
}

// The store of a join step: of the lanes of `pass`, those whose slot v lies inside the open chunk (v < CH) store; the others
// are returned (they wait for the bucket's next chunk).  exec = pass, then v_cmpx narrows it to the lanes inside -- the compare
// writes exec itself, so the whole selection costs three scalar instructions.  From code that runs with ALL lanes enabled.
__device__ __forceinline__ uint64_t pool_store_inside(uint2* wave_base, uint32_t byte_off, uint2 val, uint32_t v, uint64_t pass, uint32_t CH)
{
    const unsigned long long data = ((unsigned long long)val.y << 32) | val.x;
    uint64_t outside;
    IPK_ASSERT_FULL_EXEC();
    asm volatile("s_mov_b64 exec, %5\n\t"
                 "v_cmpx_gt_u32_e32 vcc, %6, %4\n\t"
                 "global_store_dwordx2 %1, %2, %3\n\t"
                 "s_andn2_b64 %0, %5, exec\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&s"(outside) : "v"(byte_off), "v"(data), "s"(wave_base), "v"(v), "s"(pass), "s"(CH) : "memory", "vcc");
    return outside;
}

constexpr uint32_t QUAD_SPAN_BYTES_LOG2 = 32;       // a wavefront's open chunks stay within 4 GiB of its base chunk
constexpr uint32_t QUAD_SPAN_TEST = 8;            // debug_flags bit 3: rebase every 8 chunks (tests of the rebasing path)

// floor(x / n) for 0 <= x < 128, 1 <= n <= 64, with rn = 1 / n to within a few ulp: (x + 0.5) / n is at least 1 / 128 away from every
// integer, far more than the rounding error of the product
__device__ __forceinline__ uint32_t div_small(float x_plus_half, float rn) { return (uint32_t)(x_plus_half * rn); }

// Per-wave pair appender with one {fill, where} 64-bit word per key bucket: low word = pairs reserved in the bucket's open
// chunk, high word = the chunk's byte offset from the wave's base chunk (NONE: no open chunk).  The initial state {CH, NONE}
// makes the first reservation roll; a fill >= CH never stores, so NONE is never used as an offset.
//
// base_id: chunk ids are drawn from one global counter, so a wave's open chunks are spread over everything allocated during
// its life.  When a new chunk lies 4 GiB or more past the base (or the pool is exhausted: then the "chunk" is the spare one at
// id pool_cap, which absorbs the stores of a launch that is going to be repeated with a bigger pool), every open chunk is
// closed as it is -- a descriptor may hold any count <= CH -- and the new chunk becomes the base.
template <uint32_t NB, uint32_t CH>
struct RowAppender {
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    static constexpr uint32_t CHUNK_BYTES = CH * 8u;
    static constexpr uint32_t SPAN_CHUNKS = (uint32_t)((1ull << QUAD_SPAN_BYTES_LOG2) / CHUNK_BYTES);
    const StreamParams& p;
    unsigned long long* st;                 // [NB]
    uint32_t g;
    uint32_t chunk_next = 0, chunk_end = 0;
    uint32_t base_id = 0;
    uint2* base = nullptr;                  // p.pool + base_id * CH

    // wave_gid: this wavefront's number in the launch.  With pre-assigned first chunks (p.pre_chunks; the host grants them only while
    // the whole pool lies within one store window of 4 GiB, so that no rebase can follow) bucket b starts, empty, in chunk
    // wave_gid * NB + b and that chunk is the base: a wavefront's first touch of each of its buckets is then no chunk roll --
    // 40 % of all rolls at cfg2, more for a rank's share.
    __device__ __forceinline__ void init(uint32_t wave_gid)
    {
        const uint32_t first = wave_gid * NB;
        const bool pre = p.pre_chunks != 0 && first + NB <= p.pre_chunks && first + NB <= p.pool_cap;
        for (uint32_t b = lane_id(); b < NB; b += 64)
            st[b] = pre ? ((unsigned long long)(b * CHUNK_BYTES) << 32) : (((unsigned long long)NONE << 32) | (unsigned long long)CH);
        base = p.pool;
        if (pre) {
            base_id = first;
            const unsigned long long a = (unsigned long long)(p.pool + (size_t)first * CH);
            base = reinterpret_cast<uint2*>(((unsigned long long)to_sgpr((uint32_t)(a >> 32)) << 32) | to_sgpr((uint32_t)a));
        }
    }
    __device__ __forceinline__ void write_desc(uint32_t b, unsigned long long s)
    {
        const uint32_t where = (uint32_t)(s >> 32);
        if (where == NONE) return;
        const uint32_t id = base_id + where / CHUNK_BYTES;
        if (id < p.pool_cap) p.desc[id] = ((unsigned long long)(g * NB + b) << 32) | (unsigned long long)min((uint32_t)s, CH);
    }
    __device__ __forceinline__ void rebase(uint32_t nid)
    {
        wave_lds_sync();
        for (uint32_t b = lane_id(); b < NB; b += 64) {
            const unsigned long long s = st[b];
            write_desc(b, s);
            st[b] = ((unsigned long long)NONE << 32) | (unsigned long long)max((uint32_t)s, CH);   // (a fill past CH: pairs waiting for the bucket's roll)
        }
        wave_lds_sync();
        base_id = nid;
        const unsigned long long a = (unsigned long long)(p.pool + (size_t)nid * CH);
        base = reinterpret_cast<uint2*>(((unsigned long long)to_sgpr((uint32_t)(a >> 32)) << 32) | to_sgpr((uint32_t)a));
    }
    // bucket b's chunk is full: close it, open a new one; returns the new chunk's byte offset from `base`
    __device__ __forceinline__ uint32_t roll(uint32_t b)
    {
        const uint32_t lane = lane_id();
        if (chunk_next == chunk_end) {
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(p.pool_next, ALLOC_BATCH);
            chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            chunk_end = chunk_next + ALLOC_BATCH;
        }
        uint32_t nid = to_sgpr(chunk_next);                             // (opaque: otherwise the ids live in VGPRs, and so would `base`)
        chunk_next = nid + 1;
        if (nid >= p.pool_cap && lane == 0) atomicOr(p.pool_ovf, 1u);
        nid = min(nid, p.pool_cap);                                     // (not assigned under the branch: a value merged after a divergent
                                                                        //  branch counts as divergent, and the whole state would move to VGPRs)
        if (nid - base_id >= ((p.flags & 8u) ? QUAD_SPAN_TEST : SPAN_CHUNKS)) rebase(nid);
        const uint32_t where = (nid - base_id) * CHUNK_BYTES;
        wave_lds_sync();
        const unsigned long long old = st[b];
        wave_lds_sync();
        if (lane == 0) {
            write_desc(b, old);                                            // (full: its fill is >= CH)
            st[b] = ((unsigned long long)where << 32) | (unsigned long long)((uint32_t)old - CH);
        }
        wave_lds_sync();
        return where;
    }
    __device__ __forceinline__ void close()
    {
        wave_lds_sync();
        for (uint32_t b = lane_id(); b < NB; b += 64) write_desc(b, st[b]);
    }
};

// The appender of the row-per-lane final join (ROWLANE, k = 11, 12): one {reserved pairs, chunk id} word pair per key bucket.
// Every lane (= one row of L) reserves its whole run of pairs with ONE 64-bit LDS atomic, and a run never straddles two
// chunks: when a bucket's reservations pass the chunk's end, roll_at() closes the chunk at the first run that does not fit
// (a descriptor may hold any count <= CH) and that run and the ones behind it start a new chunk.  Lanes address the pool
// with their own 64-bit pointers, so chunk ids need no common base (no RowAppender::rebase).
template <uint32_t NB, uint32_t CH>
struct LaneAppender {
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    const StreamParams& p;
    unsigned long long* st;                 // [NB]  low word = pairs reserved in the open chunk, high word = its id
    uint32_t g;
    uint32_t chunk_next = 0, chunk_end = 0;

    // wave_gid: this wavefront's number in the launch.  With pre-assigned first chunks (p.pre_chunks) bucket b starts, empty, in chunk
    // wave_gid * NB + b: at k = 12 a wavefront touches its 512 buckets once each early on, and drawing those first chunks one roll
    // at a time was more than half of all the rolls of a 125-group share (4.6 M first touches against 3.9 M chunks filled).
    __device__ __forceinline__ void init(uint32_t wave_gid)
    {
        const uint32_t first = wave_gid * NB;
        const bool pre = p.pre_chunks != 0 && first + NB <= p.pre_chunks && first + NB <= p.pool_cap;
        for (uint32_t b = lane_id(); b < NB; b += 64)
            st[b] = pre ? ((unsigned long long)(first + b) << 32) : (((unsigned long long)NONE << 32) | (unsigned long long)CH);   // NONE: "full", the first run rolls
    }
    // Bucket b's open chunk ends at x pairs (x <= CH: the start of the first run that does not fit); the runs from x on move to
    // a new chunk, whose id is returned (should they exceed that chunk too, the caller finds the next crossing run there).
    __device__ __forceinline__ uint32_t roll_at(uint32_t b, uint32_t x)
    {
        const uint32_t lane = lane_id();
        if (chunk_next == chunk_end) {
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(p.pool_next, ALLOC_BATCH);
            chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            chunk_end = chunk_next + ALLOC_BATCH;
        }
        uint32_t nid = to_sgpr(chunk_next);
        chunk_next = nid + 1;
        if (nid >= p.pool_cap && lane == 0) atomicOr(p.pool_ovf, 1u);
        nid = min(nid, p.pool_cap);                                     // (exhausted pool: the spare chunk absorbs the stores; the launch is redone)
        wave_lds_sync();
        const unsigned long long old = st[b];
        wave_lds_sync();
        if (lane == 0) {
            const uint32_t oid = (uint32_t)(old >> 32);
            if (oid < p.pool_cap) p.desc[oid] = ((unsigned long long)(g * NB + b) << 32) | (unsigned long long)x;
            st[b] = ((unsigned long long)nid << 32) | (unsigned long long)((uint32_t)old - x);
        }
        wave_lds_sync();
        return nid;
    }
    __device__ __forceinline__ void close()
    {
        wave_lds_sync();
        for (uint32_t b = lane_id(); b < NB; b += 64) {
            const unsigned long long s = st[b];
            const uint32_t id = (uint32_t)(s >> 32);
            if (id < p.pool_cap) p.desc[id] = ((unsigned long long)(g * NB + b) << 32) | (unsigned long long)min((uint32_t)s, CH);
        }
    }
};

// One column of the row-per-lane join's second pass: the lanes with s > eps (pk_compute.cpp:91) store their pair (x, s) at
// their row's cursor and advance it by one pair.  The compare writes exec itself, so test + store + advance + restore are four
// instructions (a compare into a scalar mask, exec written from it, an add-with-carry on a pair counter and the address
// arithmetic were nine).  From code that runs with ALL lanes enabled.
__device__ __forceinline__ void store_pair_if_gt(unsigned long long& cursor, uint32_t x, float s, float eps)
{
    u32x2_t data; data.x = x; data.y = __float_as_uint(s);
    IPK_ASSERT_FULL_EXEC();
    // (no "memory" clobber: nothing in these kernels reads the pool back, and the clobber would pin the LDS reads around it)
    asm volatile("v_cmpx_lt_f32_e32 vcc, %3, %2\n\t"
                 "global_store_dwordx2 %0, %1, off\n\t"
                 "v_lshl_add_u64 %0, %0, 0, 8\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(cursor) : "v"(data), "v"(s), "s"(eps) : "vcc");
}

// COUNT_ONLY: the pool-sizing pre-pass of a context's first call -- same windows, same lists, the final join only counts its
// survivors (nothing is reserved or stored); a separate instantiation so that it shows under its own name in a kernel trace.
//
// ROWLANE (k = 11, 12): the final join with one row of L per LANE.  At k = 12 a window's half lists hold ~49 x ~43 entries and a
// quarter of the candidate pairs pass: the candidate-per-lane join above spends ~49 steps of ~26 instructions per window, each
// with its own LDS atomic + ds_bpermute round trip, at 9 wavefronts per CU (nothing hides the latency).  Here lane i keeps row
// a_i; a block of <= CB entries of R is walked by COLUMN -- b_j is a broadcast LDS read (every lane the same address), one
// add + compare + add-with-carry per column counts the row's passing pairs (pass 1); every row reserves its run with one
// LDS atomic (one round trip per (row block, column block), not per step); pass 2 walks the columns again and the passing
// lanes store to their run, `row pointer + 8 * pairs so far`.  Same float operations, same pairs (pk_compute.cpp:90-91).
//
// ONEWIN (k = 11, 12, with ROWLANE): the child nodes of ONE window per wavefront step, its four nodes in the four 16-lane slots,
// instead of two nodes of two windows: the same number of steps per window at k = 12 (all four nodes have three symbols), but
// only one window's child lists are alive at a time -- 2 KB less LDS per wavefront, which with 32-window tiles and half-list
// capacity 384 admits a fourth workgroup per CU (12 wavefronts instead of 9).
template <int SIGMA, int K, int CAP, int TW, int NW, uint32_t TBL, bool COUNT_ONLY = false, bool ROWLANE = false, bool ONEWIN = false>
__global__ __launch_bounds__(NW * 64) void score_quad_kernel(StreamParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    using Q = QuadGeo<SIGMA, K>;
    using QT = QuadTile<SIGMA, K, TW>;
    static_assert(Q::OK, "quad kernel: DNA k = 8..12");
    constexpr uint32_t T = ipow(SIGMA, K);
    constexpr uint32_t NB = (T + TBL - 1) / TBL;
    constexpr uint32_t CH = chunk_pairs<TBL>();
    constexpr uint32_t mulR = Q::mulR;
    static_assert(TBL % mulR == 0, "a row of the final join must stay inside one bucket");
    constexpr uint32_t CAPL = Geo<SIGMA, Q::HL, CAP>::CAPH, CAPR = Geo<SIGMA, Q::HR, CAP>::CAPH;
    constexpr uint32_t WS = quad_wave_entries<SIGMA, K, CAP, ONEWIN>();
    constexpr uint32_t WPW = ONEWIN ? 1u : 2u;                     // windows per wavefront step
    constexpr int NST = ONEWIN ? Q::NS1 : Q::NSTEPS;
    float* cols = reinterpret_cast<float*>(smem);
    float* best = cols + QT::COLS_F;
    float* thr = best + QT::BEST_F;
    uint2* scratch_all = reinterpret_cast<uint2*>(smem + QT::HEAD_BYTES);
    unsigned long long* state_all = reinterpret_cast<unsigned long long*>(scratch_all + (size_t)NW * WS);

    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g = blockIdx.x / p.S, seg = blockIdx.x - g * p.S;
    const uint32_t m0 = p.gm_off[g], nm = p.gm_off[g + 1] - m0;
    const uint32_t total_tiles = nm * p.tiles_per_mat;
    const uint32_t t_lo = (uint32_t)(((uint64_t)total_tiles * seg) / p.S);
    const uint32_t t_hi = (uint32_t)(((uint64_t)total_tiles * (seg + 1)) / p.S);

    uint2* child = scratch_all + (size_t)wave * WS;                // [WPW][CW]
    uint2* lp = child + WPW * Q::CW;                               // L list (codes already multiplied by mulR)
    uint2* rp = lp + CAPL;                                         // R list
    using Appender = std::conditional_t<ROWLANE, LaneAppender<NB, CH>, RowAppender<NB, CH>>;
    Appender app{p, state_all + (size_t)wave * NB, g};
    app.init(blockIdx.x * NW + wave);
    unsigned long long emitted = 0;

    // lane roles: slot = lane / 16 = (window half, node slot of the step), 16 candidates per slot
    const uint32_t l16 = lane & 15u, slot = lane >> 4, half = ONEWIN ? 0u : lane >> 5, sb = slot & 1u;
    const uint32_t lt16 = (1u << l16) - 1u;
    const float lane_h = (float)lane + 0.5f;
    const float eps = p.eps;
    constexpr bool count_only = COUNT_ONLY;
    // per step: the node this lane's slot evaluates
    uint32_t st_j4[NST], st_th[NST], st_off[NST], st_cmul[NST], st_code[NST];
    bool st_on[NST];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        constexpr QuadNode none{0, 2, 0, 0, 0, 1u};
        if constexpr (ONEWIN) {
            QuadNode nd = none;
            bool on = false;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int n = Q::step1_node(s, sl);
                if (n >= 0 && slot == (uint32_t)sl) { nd = Q::node(n); on = true; }
            }
            st_j4[s] = (uint32_t)nd.J * 4u; st_th[s] = (uint32_t)nd.TH; st_off[s] = (uint32_t)nd.OFF; st_cmul[s] = nd.CMUL;
            st_code[s] = l16 * st_cmul[s]; st_on[s] = on;
        } else {
            const QuadStep qs = Q::step(s);
            const QuadNode na = Q::node(qs.A), nb = qs.B >= 0 ? Q::node(qs.B) : none;
            st_j4[s] = sb ? (uint32_t)nb.J * 4u : (uint32_t)na.J * 4u;
            st_th[s] = sb ? (uint32_t)nb.TH : (uint32_t)na.TH;
            st_off[s] = (sb ? (uint32_t)nb.OFF : (uint32_t)na.OFF) + half * Q::CW;
            st_cmul[s] = sb ? nb.CMUL : na.CMUL;
            st_code[s] = l16 * st_cmul[s];
            st_on[s] = sb ? qs.B >= 0 : true;
        }
    }

    // Tiles: a fixed range per workgroup, or -- p.tile_next -- drawn one at a time from the group's counter, so that the group's S
    // workgroups finish together whatever their tiles held (a launch of ONE round of resident workgroups, a rank's share of a
    // strong-scaled build, otherwise lasts as long as its slowest workgroup).  The next tile's number is drawn while this one is
    // staged: thread 0 issues the atomic ahead of its staging loads and parks the result in LDS behind them.
    __shared__ uint32_t sh_tile;
    const bool draw = p.tile_next != nullptr;
    uint32_t t = t_lo, t_end = t_hi;
    if (draw) { t = seg; t_end = total_tiles; }            // (the first tile by position -- the counters start at S: no burst of atomics at launch)
    while (t < t_end) {
        const uint32_t q = t / p.tiles_per_mat, tile = t - q * p.tiles_per_mat;
        const uint32_t mat = p.gm_list[m0 + q];
        const uint32_t t0 = tile * TW;
        const uint32_t nw = min((uint32_t)TW, p.nwin - t0);
        const uint32_t ncol = nw + K - 1;
        __syncthreads();                                   // previous tile fully consumed (and its sh_tile read by everyone)
        uint32_t t_drawn = 0;
        if (draw && threadIdx.x == 0) t_drawn = atomicAdd(&p.tile_next[g], 1u);
        {
            const float4* src = reinterpret_cast<const float4*>(p.logp + ((size_t)mat * p.sites + t0) * SIGMA);
            float4* dst = reinterpret_cast<float4*>(cols);
            const uint32_t n4 = ncol * (SIGMA / 4);
            for (uint32_t i = threadIdx.x; i < n4; i += NW * 64) dst[i] = src[i];
            const float* bsrc = p.best + (size_t)mat * (p.sites + 1) + t0;
            for (uint32_t i = threadIdx.x; i <= ncol; i += NW * 64) best[i] = bsrc[i];
        }
        if (draw && threadIdx.x == 0) sh_tile = t_drawn;
        __syncthreads();
        // the tile's thresholds: one thread per window
        for (uint32_t w = threadIdx.x; w < nw; w += NW * 64) {
            const float* bs = best + w;
            float* o = thr + w * Q::TH_F;
            const float eps_l = eps - (bs[K] - bs[Q::HL]);                      // pk_compute.cpp:54 at (0, K)
            const float eps_r = eps - (bs[Q::HL] - bs[0]);                      // :55
            o[0] = eps_l; o[1] = eps_r;
            quad_node_thresholds<0, Q::LA>(bs, eps_l - (bs[Q::HL] - bs[Q::LA]), o + Q::node(0).TH);             // :54 at (0, HL)
            quad_node_thresholds<Q::LA, Q::LB>(bs, eps_l - (bs[Q::LA] - bs[0]), o + Q::node(1).TH);             // :55
            quad_node_thresholds<Q::HL, Q::RA>(bs, eps_r - (bs[K] - bs[Q::HL + Q::RA]), o + Q::node(2).TH);     // :54 at (HL, HR)
            quad_node_thresholds<Q::HL + Q::RA, Q::RB>(bs, eps_r - (bs[Q::HL + Q::RA] - bs[Q::HL]), o + Q::node(3).TH);   // :55
        }
        __syncthreads();

        const uint32_t npair = (nw + WPW - 1) / WPW;
        const float* thw = thr + __umul24(wave * WPW + half, (uint32_t)Q::TH_F);    // this lane's window's thresholds: advanced, not recomputed
        for (uint32_t pr = wave; pr < npair; pr += NW, thw += WPW * NW * Q::TH_F) {
            const uint32_t wq = pr * WPW;
            const uint32_t wl = wq + half;                                       // this lane's window
            const bool wvalid = wl < nw;
            uint32_t counts[3] = {0, 0, 0};                                      // per step: list length of this lane's slot
            wave_lds_sync();                                                     // the previous pair's lists are consumed
            auto run_step = [&](auto S) {
                constexpr int s = decltype(S)::value;
                if constexpr (s < NST) {
                    const bool valid = wvalid && st_on[s];
                    const uint64_t vmask = ballot64(valid);
                    const uint32_t base = wl * 4 + st_j4[s];
                    constexpr int H = ONEWIN ? Q::step1_h(s) : Q::step(s < Q::NSTEPS ? s : 0).H;
                    quad_step<H>(cols, base, valid, vmask, thw + st_th[s], child + st_off[s], l16, slot, lt16, st_code[s], st_cmul[s], counts[s]);
                }
            };
            run_step(std::integral_constant<int, 0>{});
            run_step(std::integral_constant<int, 1>{});
            run_step(std::integral_constant<int, 2>{});
            wave_lds_sync();
            const uint32_t nwp = min(WPW, nw - wq);

#pragma unroll 1
            for (uint32_t gw = 0; gw < nwp; ++gw) {
                const uint32_t w = wq + gw;
                // list lengths: lane (window gw, node slot) * 16 holds them
                constexpr int so0 = ONEWIN ? Q::step1_of(0) : Q::step_of(0), so1 = ONEWIN ? Q::step1_of(1) : Q::step_of(1);
                constexpr int so2 = ONEWIN ? Q::step1_of(2) : Q::step_of(2), so3 = ONEWIN ? Q::step1_of(3) : Q::step_of(3);
                constexpr int sl0 = ONEWIN ? Q::slot1_of(0) : Q::slot_of(0), sl1 = ONEWIN ? Q::slot1_of(1) : Q::slot_of(1);
                constexpr int sl2 = ONEWIN ? Q::slot1_of(2) : Q::slot_of(2), sl3 = ONEWIN ? Q::slot1_of(3) : Q::slot_of(3);
                const uint32_t wbase = ONEWIN ? 0u : gw * 32u;
                const uint32_t nla = (uint32_t)__builtin_amdgcn_readlane((int)counts[so0], (int)(wbase + sl0 * 16));
                const uint32_t nlb = (uint32_t)__builtin_amdgcn_readlane((int)counts[so1], (int)(wbase + sl1 * 16));
                const uint32_t nra = (uint32_t)__builtin_amdgcn_readlane((int)counts[so2], (int)(wbase + sl2 * 16));
                const uint32_t nrb = (uint32_t)__builtin_amdgcn_readlane((int)counts[so3], (int)(wbase + sl3 * 16));
                if (nla * nlb * nra * nrb == 0) continue;                        // an empty half: nothing survives  (each <= 64)
                const uint2* cw = child + gw * Q::CW;
                const uint2 *la = cw + Q::node(0).OFF, *lb = cw + Q::node(1).OFF, *ra = cw + Q::node(2).OFF, *rb = cw + Q::node(3).OFF;
                const float eps_l = thr[w * Q::TH_F], eps_r = thr[w * Q::TH_F + 1];
                const uint32_t tl = nla * nlb, tr = nra * nrb;
                uint32_t nL, nR;
                // A half list beyond the wavefront's capacity (0.13 % of cfg2's windows) is taken in SLICES: as many rows of its first
                // child as are sure to fit (CAP / 64 at least), every slice of L against every slice of R -- the same pairs in another
                // order, through the same appender.  Until round 4 such windows were queued for a kernel of their own that put their
                // pairs into the finished tables with global atomics (0.6 ms at cfg2, 0.14 ms of a rank's 2.9-ms share).
                // Only with the row-per-lane join (k = 11, 12), whose count pass costs two instructions per column: the candidate-per-lane
                // join of k <= 10 spends ~25 per step whenever one of a trip's 192 candidates passes, and a big window's |L| x |R| / 64
                // steps through it came to 12.2 instead of 10.2 ms at cfg2 -- those windows keep their queue and their kernel.
                constexpr bool SLICE = ROWLANE;
                constexpr uint32_t SAFE_L = CAPL / Q::FLB, SAFE_R = CAPR / Q::FRB;
                static_assert(SAFE_L >= 1 && SAFE_R >= 1, "a slice of one row of the first child must fit the list");
                const bool one_step = tl <= 64 && tr <= 64;
                uint32_t sla = nla, sra = nra, ia = 0, ir = 0;                   // rows of la / ra per slice, first row of the current slices
                bool keep_l = false;                                             // L's current slice is built (R's slices pass by)
                for (;;) {
                wave_lds_sync();                                                 // the previous window's (slice's) L / R are consumed
                if (one_step) {
                    // both half joins in one step
                    const float rnl = __builtin_amdgcn_rcpf((float)nlb), rnr = __builtin_amdgcn_rcpf((float)nrb);   // (1 ulp: ample for div_small)
                    const uint32_t il = div_small(lane_h, rnl), jl = (uint32_t)(__mul24((int)il, -(int)nlb) + (int)lane);
                    const uint32_t ir = div_small(lane_h, rnr), jr = (uint32_t)(__mul24((int)ir, -(int)nrb) + (int)lane);
                    // lanes past a half's candidates read the entries that follow in LDS and are masked out of the ballots
                    const uint2 a0 = la[il], b0 = lb[jl], a1 = ra[ir], b1 = rb[jr];
                    const float s0 = __uint_as_float(a0.y) + __uint_as_float(b0.y);          // :90
                    const float s1 = __uint_as_float(a1.y) + __uint_as_float(b1.y);
                    const bool v0 = lane < tl, v1 = lane < tr, c0 = s0 > eps_l, c1 = s1 > eps_r;    // :91
                    const bool p0 = v0 && c0, p1 = v1 && c1;
                    const uint64_t h0 = ballot64(v0) & ballot64(c0), h1 = ballot64(v1) & ballot64(c1);
                    if (p0) lp[mbcnt(h0)] = make_uint2(a0.x * (Q::FLB * mulR) + b0.x, __float_as_uint(s0));   // lb codes carry mulR
                    if (p1) rp[mbcnt(h1)] = make_uint2(a1.x * Q::FRB + b1.x, __float_as_uint(s1));
                    nL = (uint32_t)__popcll(h0); nR = (uint32_t)__popcll(h1);
                } else {
                    // (the list lengths are wave-uniform by construction; without the readfirstlane the branches below count as
                    //  divergent, and every value carried around this loop -- the appender's state -- is then held in VGPRs)
                    if (!keep_l)
                        nL = (uint32_t)__builtin_amdgcn_readfirstlane((int)join_to_list(la + ia, min(sla, nla - ia), lb, nlb, eps_l, Q::FLB * mulR, lp, CAPL));
                    if constexpr (SLICE) {
                        if (nL == LIST_OVERFLOW) { sla = SAFE_L; continue; }            // (only a whole list overflows: ia == 0)
                    }
                    nR = nL == LIST_OVERFLOW || nL == 0 ? nL : (uint32_t)__builtin_amdgcn_readfirstlane((int)join_to_list(ra + ir, min(sra, nra - ir), rb, nrb, eps_r, Q::FRB, rp, CAPR));
                    if constexpr (SLICE) {
                        if (nR == LIST_OVERFLOW) { sra = SAFE_R; keep_l = true; continue; }   // (ir == 0; L's slice stays as it is)
                    } else if (nL == LIST_OVERFLOW || nR == LIST_OVERFLOW) {
                        if (lane == 0) {
                            const uint32_t qi = atomicAdd(p.ovf_count, 1u);
                            p.ovf_queue[qi] = ((unsigned long long)mat << 32) | (unsigned long long)(t0 + w);
                        }
                        // (keeps the join of the one-lane branch out of the loop latch: a phi there that also merges other paths
                        //  would count as divergent, and with it every value carried around the loop -- see RowAppender::roll)
                        __builtin_amdgcn_wave_barrier();
                        break;
                    }
                }
                if (nL != 0 && nR != 0) {
                wave_lds_sync();
                if (p.flags & 4u) emitted += nL + nR;                               // diagnostics: list building only
                else
                if constexpr (ROWLANE) {
                    // ---- final join, one row of L per lane ---------------------------------------------------------------
                    constexpr uint32_t CB = 64;                                       // columns per round: a row's run is <= 64 <= CH pairs
                    static_assert(CB <= CH, "row-per-lane join: a run must fit a chunk");
                    for (uint32_t ib = 0; ib < nL; ib += 64) {
                        const uint2 a = lp[ib + lane];                               // (rows past nL read on into R: they never pass, ay = -inf)
                        const bool rv = ib + lane < nL;
                        const float ay = rv ? __uint_as_float(a.y) : -__builtin_inff();
                        const uint32_t bk = a.x / TBL;
                        for (uint32_t jb = 0; jb < nR; jb += CB) {
                            const uint32_t nc = min(CB, nR - jb);                    // columns of this round (wave-uniform)
                            const uint2* rj = rp + jb;                               // column j: one broadcast LDS read for all rows
                            // pass 1: pairs of the row that pass                      pk_compute.cpp:90-91
                            uint32_t cnt = 0;
                            if constexpr (count_only) {
                                for (uint32_t j = 0; j < nc; ++j) emitted += (uint32_t)__popcll(ballot64(ay + __uint_as_float(rj[j].y) > eps));
                            } else {
                                // (compare + add-with-carry per column, spelled out: left to itself the compiler pairs the columns and
                                //  spends a v_cndmask per column on it)
                                auto count8 = [&](auto NC, uint32_t j0, uint32_t& n, float e) {   // (n, e: an asm operand must not be a capture)
                                    constexpr uint32_t N = decltype(NC)::value;
                                    float by[N];
#pragma unroll
                                    for (uint32_t u = 0; u < N; ++u) by[u] = __uint_as_float(rj[j0 + u].y);
#pragma unroll
                                    for (uint32_t u = 0; u < N; ++u) {
                                        const float sj = ay + by[u];                  // :90
                                        asm("v_cmp_lt_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(n) : "v"(sj), "s"(e) : "vcc");   // :91
                                    }
                                };
                                uint32_t j0 = 0;
                                for (; j0 + 8 <= nc; j0 += 8) count8(std::integral_constant<uint32_t, 8>{}, j0, cnt, eps);
                                if (j0 + 4 <= nc) { count8(std::integral_constant<uint32_t, 4>{}, j0, cnt, eps); j0 += 4; }
                                for (; j0 < nc; ++j0) count8(std::integral_constant<uint32_t, 1>{}, j0, cnt, eps);
                            }
                            if constexpr (!count_only) {
                                const bool has = cnt != 0;
                                if (ballot64(has) == 0) continue;
                                if (p.flags & 16u) { emitted += cnt; continue; }      // diagnostics: first pass only
                                // every row reserves its run: {pairs before it in the bucket's open chunk, chunk id}
                                unsigned long long got;
                                asm volatile("" : "=v"(got));
                                if (has) got = atomicAdd(&app.st[bk], (unsigned long long)cnt);
                                uint32_t pre = (uint32_t)got, id = (uint32_t)(got >> 32);
                                // the one run per bucket that reaches past the chunk's end: it and the runs behind it open a new chunk
                                // (a bucket that gained more than a chunk's worth in this round crosses again in the new chunk: re-examined)
                                for (;;) {
                                    uint64_t X = ballot64(has && pre <= CH && pre + cnt > CH);
                                    if (X == 0) break;
                                    do {
                                        const int ls = __ffsll((long long)X) - 1;
                                        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)bk, ls);
                                        const uint32_t xv = (uint32_t)__builtin_amdgcn_readlane((int)pre, ls);
                                        const uint32_t xid = (uint32_t)__builtin_amdgcn_readlane((int)id, ls);
                                        const uint32_t nid = app.roll_at(bb, xv);
                                        const bool sel = has && bk == bb && id == xid && pre >= xv;       // (id: runs left behind in an earlier chunk stay)
                                        id = sel ? nid : id;
                                        pre = sel ? pre - xv : pre;
                                        X &= X - 1ull;
                                    } while (X);
                                }
                                unsigned long long run = reinterpret_cast<unsigned long long>(p.pool) + (((unsigned long long)id * CH + pre) << 3);
                                // pass 2: the same columns again; the passing lanes store to their run
                                if (p.flags & 2u) continue;                          // diagnostics: no second pass
                                // (eight columns' reads are issued together: one LDS round trip per eight columns, not per column)
                                auto emit8 = [&](auto NC, uint32_t j0) {
                                    constexpr uint32_t N = decltype(NC)::value;
                                    uint2 b[N];
#pragma unroll
                                    for (uint32_t u = 0; u < N; ++u) b[u] = rj[j0 + u];
#pragma unroll
                                    for (uint32_t u = 0; u < N; ++u)
                                        store_pair_if_gt(run, a.x + b[u].x, ay + __uint_as_float(b[u].y), eps);   // :90, :91
                                };
                                uint32_t j0 = 0;
                                for (; j0 + 8 <= nc; j0 += 8) emit8(std::integral_constant<uint32_t, 8>{}, j0);
                                if (j0 + 4 <= nc) { emit8(std::integral_constant<uint32_t, 4>{}, j0); j0 += 4; }
                                for (; j0 < nc; ++j0) emit8(std::integral_constant<uint32_t, 1>{}, j0);
                            }
                        }
                    }
                } else
                // ---- final join: a block of <= 64 entries of R in registers, floor(64 / width) rows of L per step,
                //      two steps per trip where two remain (their LDS round trips overlap) -----------------------------
                for (uint32_t jb = 0; jb < nR; jb += 64) {
                    const uint32_t nRb = min(64u, nR - jb);                          // width of this block of R
                    const float rn = __builtin_amdgcn_rcpf((float)nRb);
                    const uint32_t rps = to_sgpr(div_small(64.5f, rn));              // rows per step
                    const uint32_t rsel = div_small(lane_h, rn), rfirst = __umul24(rsel, nRb), col = lane - rfirst;
                    const bool lane_ok = rsel < rps;
                    const uint2 b = rp[jb + col];
                    const float by = __uint_as_float(b.y);
                    const bool is_head = lane_ok && col == 0;
                    const uint64_t okmask = ballot64(lane_ok);
                    const uint64_t rowbits = nRb >= 64 ? ~0ull : ((1ull << nRb) - 1ull);
                    const uint64_t rowmask = lane_ok ? (rowbits << rfirst) : 0ull;
                    const uint32_t rm_lo = (uint32_t)rowmask, rm_hi = (uint32_t)(rowmask >> 32);
                    const int head_addr = (int)(rfirst << 2);
                    const uint2* ap = lp + rsel;                                     // rows past nL read on into the R list: masked below
                    // NS steps starting at row i0; FULL: every step has all its rows
                    auto trip = [&](auto NSC, auto FULLC, uint32_t i0) {
                        constexpr int NS = decltype(NSC)::value;
                        constexpr bool FULL = decltype(FULLC)::value;
                        uint2 a[NS]; float s[NS]; bool live[NS]; uint64_t m[NS]; uint64_t any = 0;
#pragma unroll
                        for (int u = 0; u < NS; ++u) a[u] = ap[i0 + u * rps];
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            s[u] = __uint_as_float(a[u].y) + by;                     // pk_compute.cpp:90
                            const bool c = s[u] > eps;                               // :91
                            if constexpr (FULL) { live[u] = lane_ok; m[u] = okmask & ballot64(c); }
                            else {
                                const uint32_t nv = min(rps, nL - i0 - u * rps) * nRb;   // live lanes of the step
                                live[u] = lane < nv; m[u] = ballot64(live[u]) & ballot64(c);
                            }
                            any |= m[u];
                        }
#ifndef IPK_QNOANY
                        if (any == 0) return;
#endif
                        if constexpr (count_only) {                                   // the pool-sizing pre-pass
#pragma unroll
                            for (int u = 0; u < NS; ++u) emitted += (uint32_t)__popcll(m[u]);
                            return;
                        } else {
                        uint32_t rank[NS], bk[NS], rc[NS]; unsigned long long got[NS];
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            const uint32_t xl = (uint32_t)m[u] & rm_lo, xh = (uint32_t)(m[u] >> 32) & rm_hi;
                            rank[u] = __builtin_amdgcn_mbcnt_hi(xh, __builtin_amdgcn_mbcnt_lo(xl, 0u));
                            rc[u] = (uint32_t)__popc(xl) + (uint32_t)__popc(xh);
                            bk[u] = a[u].x / TBL;
                            asm volatile("" : "=v"(got[u]));                         // only the head lanes' values are read (bpermute below)
                        }
                        if constexpr (FULL) {                                        // one exec region for the steps' reservations
                            if (is_head) {
#pragma unroll
                                for (int u = 0; u < NS; ++u) got[u] = atomicAdd(&app.st[bk[u]], (unsigned long long)rc[u]);
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < NS; ++u)
                                if (is_head && live[u]) got[u] = atomicAdd(&app.st[bk[u]], (unsigned long long)rc[u]);
                        }
                        uint32_t v[NS]; uint2 val[NS]; uint64_t ovf[NS]; uint64_t anyo = 0;
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            const uint32_t fill = (uint32_t)__builtin_amdgcn_ds_bpermute(head_addr, (int)(uint32_t)got[u]);
                            const uint32_t where = (uint32_t)__builtin_amdgcn_ds_bpermute(head_addr, (int)(uint32_t)(got[u] >> 32));
                            v[u] = fill + rank[u];
                            val[u] = make_uint2(a[u].x + b.x, __float_as_uint(s[u]));
                            ovf[u] = pool_store_inside(app.base, lshl3_add(v[u], where), val[u], v[u], m[u], CH);
                            anyo |= ovf[u];
                        }
                        while (anyo) {                                                // a bucket's chunk filled up: open a new one
                            uint32_t bb = 0;
                            bool found = false;
#pragma unroll
                            for (int u = 0; u < NS; ++u)
                                if (!found && ovf[u]) { bb = (uint32_t)__builtin_amdgcn_readlane((int)bk[u], (int)(__ffsll((long long)ovf[u]) - 1)); found = true; }
                            const uint32_t nwhere = app.roll(bb);
                            anyo = 0;
#pragma unroll
                            for (int u = 0; u < NS; ++u) {
                                const bool h = ((ovf[u] >> lane) & 1ull) != 0 && bk[u] == bb;
                                if (h) pool_store(app.base, nwhere + ((v[u] - CH) << 3), val[u]);
                                ovf[u] &= ~ballot64(h);
                                anyo |= ovf[u];
                            }
                        }
                        }
                    };
                    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                    uint32_t i0 = 0;
                    // three steps per trip while three remain (-2 % against two at cfg2: the trip's scalar bookkeeping is shared by
                    // more steps; four measured slower again)
                    using I3 = std::integral_constant<int, 3>;
                    for (; i0 + 3 * rps <= nL; i0 += 3 * rps) trip(I3{}, std::true_type{}, i0);
                    for (; i0 + 2 * rps <= nL; i0 += 2 * rps) trip(I2{}, std::true_type{}, i0);
                    if (i0 + rps < nL) trip(I2{}, std::false_type{}, i0);
                    else if (i0 < nL) trip(I1{}, std::false_type{}, i0);
                }
                }
                // the next slices: R's behind the current L, then the next slice of L with R from its start
                if (!SLICE || one_step) break;
                ir += sra;
                if (nL != 0 && ir < nra) { keep_l = true; continue; }
                ir = 0; keep_l = false; ia += sla;
                if (ia >= nla) break;
                }
            }
        }
        t = draw ? sh_tile : t + 1;                        // (sh_tile: written before the barrier behind the staging, rewritten only after the next one)
    }
    if (!count_only) app.close();
    if (lane == 0 && emitted) atomicAdd(p.emitted, emitted);       // (flags 2 / 4 only: otherwise the pairs are counted from the chunk descriptors)
}

}  // namespace ipkgpu
