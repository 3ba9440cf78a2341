// dcla_device.hpp -- wavefront-wide divide-and-conquer k-mer scoring for gfx950 (wave64).
//
// Device-side building blocks shared by the scoring kernels.  Semantics follow
// ipk/src/pk_compute.cpp:42-114 (DCLA::DC) as the set definition of SURVEY.md App. A.2:
//
//   S(j,1,e) = {(i, m[j][i]) : m[j][i] > e}
//   S(j,h,e) = {(a.key*sigma^hr + b.key, a.score + b.score)
//                 : a in S(j,hl,e - M(j+hl,hr)), b in S(j+hl,hr,e - M(j,hl)), a.score + b.score > e}
//   M(p,len) = best[p+len] - best[p]           hl = h/2, hr = h - h/2
//
// All arithmetic is binary32, one rounding per operation (build with -ffp-contract=off), every
// comparison strict, every intermediate bound applied -- the hierarchical float bounds decide
// boundary k-mers, so none of them may be skipped or re-associated.
//
// In-kernel k-mer codes are DENSE base-sigma indices (for DNA identical to the 2-bit packed
// code); the compaction kernel converts them to IPK's bit-packed codes (pk_compute.cpp:96-104).
//
// One wavefront owns one window at a time: a node's sigma^h candidates (h small) are evaluated
// one per lane; larger nodes are the filtered cross product of their children's lists, lanes
// striding the flattened candidate space, survivors compacted with ballot + mbcnt into LDS.
// Lists come out in ascending code order (lane order, then (left, right) order of the joins) --
// the pair appender of the stream variant relies on that for its bucket runs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ipkgpu {

constexpr __host__ __device__ uint32_t ipow(uint32_t b, int e) { return e <= 0 ? 1u : b * ipow(b, e - 1); }
constexpr __host__ __device__ uint32_t umin_c(uint32_t a, uint32_t b) { return a < b ? a : b; }
constexpr __host__ __device__ uint32_t umax_c(uint32_t a, uint32_t b) { return a > b ? a : b; }

constexpr uint32_t LIST_OVERFLOW = 0xFFFFFFFFu;

// Order-preserving float -> u32 map (all finite/inf values; NaN never reaches it): bigger float
// <=> bigger code; 0 is free as the "empty slot" sentinel (it decodes to a NaN pattern).
__host__ __device__ __forceinline__ uint32_t enc_score_bits(uint32_t u)
{
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ uint32_t dec_score_bits(uint32_t e)
{
    return (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// llvm.amdgcn.ballot on the i1 itself: HIP's __ballot() goes through an int and costs two extra VALU instructions
// (v_cndmask + v_cmp) wherever the predicate is a combination of compares
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// A wave-uniform value that the compiler holds in a VGPR (the result of float VALU arithmetic on uniform inputs), moved
// to an SGPR.  __builtin_amdgcn_readfirstlane is folded away when its argument is known to be uniform, which leaves loop
// counters derived from it in VGPRs (exec-masked loops, quarter-rate v_mul_lo_u32); the asm form is opaque.
__device__ __forceinline__ uint32_t to_sgpr(uint32_t x)
{
    uint32_t r;
    asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(x));
    return r;
}

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// The exec-writing asm helpers below (store8_lanes, pool_store_lanes, pool_store_inside, km_write_c's LDS scatter) end with
// `s_mov_b64 exec, -1`: they are only correct from code that runs with all 64 lanes enabled.  That holds by construction
// (uniform control flow in full wavefronts) but nothing in the language says so -- a compiler that structurises one of the
// surrounding loops as divergent would break them silently.  The IPK_EXEC_ASSERT build (ipk_amd/build.py: variant
// "execassert", run by tests/test_gpu_parity.py::test_exec_assert_build) counts every call that is entered with a partial
// exec mask; ipkgpu_debug_exec_violations() reads the count.  The shipped build compiles the check away.
#ifdef IPK_EXEC_ASSERT
__device__ unsigned int g_exec_violations;
#define IPK_ASSERT_FULL_EXEC() do { if (__builtin_amdgcn_ballot_w64(true) != ~0ull) atomicAdd(&g_exec_violations, 1u); } while (0)
#else
#define IPK_ASSERT_FULL_EXEC() ((void)0)
#endif

// One 8-byte record per lane of `mask` to uniform_base + byte_off (+ IMM bytes): the store form with a scalar base and a 32-bit
// lane offset, the lanes selected by writing exec directly -- from code that runs with ALL lanes enabled (uniform control flow
// in full wavefronts).  Two scalar instructions around the store instead of compare / and-saveexec / branch / restore.
template <int IMM = 0>
__device__ __forceinline__ void store8_lanes(const void* uniform_base, uint32_t byte_off, uint32_t x, uint32_t y, uint64_t mask)
{
    u32x2_t data; data.x = x; data.y = y;
    IPK_ASSERT_FULL_EXEC();
    asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, %2 offset:%4\n\ts_mov_b64 exec, -1"
                 : : "v"(byte_off), "v"(data), "s"(uniform_base), "s"(mask), "n"(IMM) : "memory");
}

__device__ __forceinline__ uint32_t mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Orders this wave's LDS writes before its later LDS reads of other lanes' data.  DS operations
// of one wave execute in issue order; the fences only stop the compiler from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A window as the wave sees it: LDS-resident tile of columns and the matching slice of the
// prefix array (matrix::_best_scores, window.cpp:16-27); `w` = window start within the tile.
struct WinCtx {
    const float* cols;   // [tile_cols][SIGMA]
    const float* best;   // [tile_cols + 1]
    uint32_t w;
};

// ---- direct evaluation of a small node: candidate `cand` in [0, SIGMA^H) ------------------
template <int SIGMA, int J, int H>
struct Direct {
    static __device__ __forceinline__ bool eval(const WinCtx& c, float eps, uint32_t cand, float& score)
    {
        if constexpr (H == 1) {
            score = c.cols[(c.w + J) * SIGMA + cand];                 // as_column, pk_compute.cpp:14-26
            return score > eps;
        } else {
            constexpr int HL = H / 2, HR = H - HL;
            const float eps_l = eps - (c.best[c.w + J + H] - c.best[c.w + J + HL]);   // :54
            const float eps_r = eps - (c.best[c.w + J + HL] - c.best[c.w + J]);       // :55
            constexpr uint32_t PR = ipow(SIGMA, HR);
            const uint32_t cl = cand / PR, cr = cand - cl * PR;
            float sl, sr;
            const bool okl = Direct<SIGMA, J, HL>::eval(c, eps_l, cl, sl);
            const bool okr = Direct<SIGMA, J + HL, HR>::eval(c, eps_r, cr, sr);
            score = sl + sr;                                          // :90
            return okl & okr & (score > eps);                         // :91 (strict)
        }
    }
};

// ---- list geometry (entries of 8 bytes: .x = dense code, .y = score bits) -----------------
// A node's result occupies out[0 .. CAPH); while it is being built its children live right
// after it: left child at out + CAPH, right child after the left child's result.  SCRATCH is
// what a node needs past its own result region.
template <int SIGMA, int H, int CAP>
struct Geo {
    static constexpr uint32_t FULL = ipow(SIGMA, H);
    static constexpr uint32_t CAPH = umin_c(FULL, (uint32_t)CAP);
    static constexpr bool DIRECT = FULL <= 64;
    static constexpr int HL = H / 2, HR = H - H / 2;
    static constexpr uint32_t scratch()
    {
        if constexpr (DIRECT) return 0;
        else {
            using GL = Geo<SIGMA, HL, CAP>;
            using GR = Geo<SIGMA, HR, CAP>;
            return umax_c(GL::CAPH + GL::scratch(), GL::CAPH + GR::CAPH + GR::scratch());
        }
    }
    static constexpr uint32_t SCRATCH = scratch();
};

// LDS entries one wave needs for windows of length K: the two half lists and their scratch
// (the top-level node has no result region of its own -- its survivors are emitted).
template <int SIGMA, int K, int CAP>
constexpr uint32_t wave_scratch_entries() { return Geo<SIGMA, K, CAP>::SCRATCH; }

// Exact floor(idx / d) for idx < 2^23, d >= 1: float reciprocal estimate + one-step fixup.
__device__ __forceinline__ void divmod_u(uint32_t idx, uint32_t d, float rcp_d, uint32_t& q, uint32_t& r)
{
    q = (uint32_t)(((float)idx + 0.5f) * rcp_d);
    int32_t rem = (int32_t)idx - (int32_t)(q * d);
    if (rem < 0) { q -= 1; rem += (int32_t)d; }
    else if (rem >= (int32_t)d) { q += 1; rem -= (int32_t)d; }
    r = (uint32_t)rem;
}

// Calls f(valid, a, b) wave-uniformly for every pair (a in L, b in R), 64 pairs per call.
// Short R: lanes stride the flattened nL*nR space (< 2^23 pairs: nL <= 8000, nR < 64);
// long R: one L entry at a time, lanes stride R.
template <class F>
__device__ __forceinline__ void for_each_pair(const uint2* L, uint32_t nL, const uint2* R, uint32_t nR, F&& f)
{
    const uint32_t lane = lane_id();
    if (nR < 64) {
        const uint32_t total = nL * nR;
        const float rcp = 1.0f / (float)nR;
        for (uint32_t base = 0; base < total; base += 64) {
            const uint32_t idx = base + lane;
            const bool valid = idx < total;
            uint2 a = make_uint2(0, 0), b = make_uint2(0, 0);
            if (valid) {
                uint32_t i, j;
                divmod_u(idx, nR, rcp, i, j);
                a = L[i]; b = R[j];
            }
            f(valid, a, b);
        }
    } else {
        for (uint32_t i = 0; i < nL; ++i) {
            const uint2 a = L[i];
            for (uint32_t jb = 0; jb < nR; jb += 64) {
                const uint32_t j = jb + lane;
                const bool valid = j < nR;
                uint2 b = make_uint2(0, 0);
                if (valid) b = R[j];
                f(valid, a, b);
            }
        }
    }
}

// Filtered cross product of two LDS lists into an LDS list; returns the survivor count, or
// LIST_OVERFLOW when more than `cap` survive (the window is then redone by the big-list kernel).
__device__ __forceinline__ uint32_t join_to_list(const uint2* L, uint32_t nL, const uint2* R, uint32_t nR,
                                                 float eps, uint32_t mulR, uint2* out, uint32_t cap)
{
    uint32_t n_out = 0;
    bool over = false;
    for_each_pair(L, nL, R, nR, [&](bool valid, uint2 a, uint2 b) {
        const float s = __uint_as_float(a.y) + __uint_as_float(b.y);      // pk_compute.cpp:90
        const bool pass = valid && (s > eps);                              // :91
        const uint64_t m = __ballot(pass);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (n_out + cnt > cap) { over = true; }
        else if (pass) out[n_out + mbcnt(m)] = make_uint2(a.x * mulR + b.x, __float_as_uint(s));
        n_out += cnt;
    });
    return over ? LIST_OVERFLOW : n_out;
}

// Builds S(J, H, eps) of the window into out[0 .. CAPH); children/scratch follow at out + CAPH.
template <int SIGMA, int J, int H, int CAP>
struct Node {
    using G = Geo<SIGMA, H, CAP>;
    static __device__ __forceinline__ uint32_t build(const WinCtx& c, float eps, uint2* out)
    {
        if constexpr (G::DIRECT) {
            const uint32_t lane = lane_id();
            float s = 0.f;
            bool pass = false;
            if (lane < G::FULL) pass = Direct<SIGMA, J, H>::eval(c, eps, lane, s);
            const uint64_t m = __ballot(pass);
            if (pass) out[mbcnt(m)] = make_uint2(lane, __float_as_uint(s));
            return (uint32_t)__popcll(m);                   // FULL <= 64: cannot overflow
        } else {
            constexpr int HL = G::HL, HR = G::HR;
            using GL = Geo<SIGMA, HL, CAP>;
            const float eps_l = eps - (c.best[c.w + J + H] - c.best[c.w + J + HL]);   // :54
            const float eps_r = eps - (c.best[c.w + J + HL] - c.best[c.w + J]);       // :55
            uint2* lp = out + G::CAPH;
            uint2* rp = lp + GL::CAPH;
            const uint32_t nl = Node<SIGMA, J, HL, CAP>::build(c, eps_l, lp);
            if (nl == LIST_OVERFLOW || nl == 0) return nl;
            const uint32_t nr = Node<SIGMA, J + HL, HR, CAP>::build(c, eps_r, rp);
            if (nr == LIST_OVERFLOW || nr == 0) return nr;
            wave_lds_sync();
            const uint32_t n = join_to_list(lp, nl, rp, nr, eps, ipow(SIGMA, HR), out, G::CAPH);
            wave_lds_sync();
            return n;
        }
    }
};

// Builds the two top-level half lists of a window of length K (K not DIRECT) in `scratch`
// (wave_scratch_entries<SIGMA,K,CAP>() entries): L = S(0, K/2, eps - M(right)) at scratch[0..],
// R = S(K/2, K-K/2, eps - M(left)) right after L's region.  L is built first, with its children
// over the not-yet-used R region.  Returns false when a list overflowed CAP.
template <int SIGMA, int K, int CAP>
__device__ __forceinline__ bool build_halves(const WinCtx& c, float eps, uint2* scratch,
                                             const uint2*& Lp, uint32_t& nL, const uint2*& Rp, uint32_t& nR)
{
    constexpr int HL = K / 2, HR = K - K / 2;
    using GL = Geo<SIGMA, HL, CAP>;
    const float eps_l = eps - (c.best[c.w + K] - c.best[c.w + HL]);    // pk_compute.cpp:54
    const float eps_r = eps - (c.best[c.w + HL] - c.best[c.w]);        // :55
    uint2* lp = scratch;
    uint2* rp = scratch + GL::CAPH;
    Lp = lp; Rp = rp; nL = 0; nR = 0;
    const uint32_t nl = Node<SIGMA, 0, HL, CAP>::build(c, eps_l, lp);
    if (nl == LIST_OVERFLOW) return false;
    if (nl == 0) return true;
    const uint32_t nr = Node<SIGMA, HL, HR, CAP>::build(c, eps_r, rp);
    if (nr == LIST_OVERFLOW) return false;
    nL = nl; nR = nr;
    wave_lds_sync();
    return true;
}

}  // namespace ipkgpu
