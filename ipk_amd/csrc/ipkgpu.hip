// ipkgpu.hip -- MI355X (gfx950) phylo-k-mer scoring engine: kernels + C ABI (include/ipkgpu.h).
//
// Path accelerated (IPK tree): db_builder.cpp:576-698 (explore_kmers / explore_group) ->
// window.cpp:16-27,164-182 (prefix of column maxima, sliding windows) -> pk_compute.cpp:42-114
// (DCLA::DC) -> branch_group.cpp:88-101 (put: per-branch max-reduce).
//
// Data layout in HBM:
//   logp   [n_mats][sites][sigma] f32        caller's matrices, read once per scoring pass
//   best   [n_mats][sites+1]      f32        sequential float prefix sums of column maxima
//   table  [groups_in_batch][sigma^k] u32    per-group dense max table, order-preserving score
//                                            codes, 0 = empty (the on-device group_hash_map)
//   result CSR: offsets[g], keys[] (bit-packed codes, ascending per group), scores[] f32
//
// There is NO CPU fallback: without a GPU ipkgpu_create fails with IPKGPU_ERR_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ipkgpu.h"
#include "dcla_device.hpp"

using namespace ipkgpu;

// =============================================================================================
// kernels
// =============================================================================================

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

// =============================================================================================
// host side
// =============================================================================================

struct ipkgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int64_t workspace_bytes = 0;
    int64_t opt_list_cap = 0;
    int64_t opt_variant = 0;
    // grow-only cached workspaces
    void* table = nullptr;      size_t table_cap = 0;
    void* best = nullptr;       size_t best_cap = 0;
    void* ovfq = nullptr;       size_t ovfq_cap = 0;
    void* counts = nullptr;     size_t counts_cap = 0;
    void* offsets = nullptr;    size_t offsets_cap = 0;
    void* goff = nullptr;       size_t goff_cap = 0;
    void* small = nullptr;      // emitted (u64) + ovf_count (u32)
    void* idx = nullptr;        size_t idx_cap = 0;   // mat_list + mat_slot
    int num_cu = 256;
};

struct ipkgpu_result {
    ipkgpu_ctx* ctx = nullptr;
    std::vector<uint32_t> group_ids;
    std::vector<uint64_t> offsets;
    uint64_t emitted = 0;
    uint32_t* d_keys = nullptr;
    float* d_scores = nullptr;
    size_t cap = 0;
    std::vector<uint32_t> h_keys;
    std::vector<float> h_scores;
    bool h_keys_ok = false, h_scores_ok = false;
    double t_total = 0, t_prefix = 0, t_score = 0, t_compact = 0;
    int score_launches = 0;
};

static std::string g_create_err;

static int fail(ipkgpu_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? IPKGPU_ERR_NOMEM : IPKGPU_ERR_HIP,     \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int ensure(ipkgpu_ctx* ctx, void** p, size_t* cap, size_t need)
{
    if (*cap >= need && *p) return IPKGPU_OK;
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIP_TRY(ctx, hipMalloc(p, need));
    *cap = need;
    return IPKGPU_OK;
}

extern "C" {

uint32_t ipkgpu_bits_per_symbol(uint32_t sigma) { return sigma == 4 ? 2u : sigma == 20 ? 5u : 0u; }
uint32_t ipkgpu_max_k(uint32_t sigma) { return sigma == 4 ? 12u : sigma == 20 ? 6u : 0u; }
size_t ipkgpu_kmer_batch(uint32_t key, size_t n_ranges) { return n_ranges ? key % n_ranges : 0; }

float ipkgpu_log_threshold(float omega, uint32_t sigma, uint32_t k)
{
    return log10f(powf(omega / (float)sigma, (float)k));
}

const char* ipkgpu_last_error(const ipkgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int ipkgpu_create(int device_id, ipkgpu_ctx** out)
{
    if (!out) return fail(nullptr, IPKGPU_ERR_INVALID, "ipkgpu_create: null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, IPKGPU_ERR_NODEVICE, "no HIP device available (%s); this engine has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, IPKGPU_ERR_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
    ipkgpu_ctx* ctx = new (std::nothrow) ipkgpu_ctx();
    if (!ctx) return fail(nullptr, IPKGPU_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, IPKGPU_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) ctx->workspace_bytes = (int64_t)(free_b / 2);
    else ctx->workspace_bytes = (int64_t)8 << 30;
    if ((e = hipMalloc(&ctx->small, 64)) != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(nullptr, IPKGPU_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return IPKGPU_OK;
}

void ipkgpu_destroy(ipkgpu_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    void* bufs[] = {ctx->table, ctx->best, ctx->ovfq, ctx->counts, ctx->offsets, ctx->goff, ctx->small, ctx->idx};
    for (void* b : bufs) if (b) hipFree(b);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ipkgpu_set_option(ipkgpu_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return IPKGPU_ERR_INVALID;
    if (!strcmp(name, "workspace_bytes")) {
        if (value <= 0) return fail(ctx, IPKGPU_ERR_INVALID, "workspace_bytes must be positive");
        ctx->workspace_bytes = value;
        return IPKGPU_OK;
    }
    if (!strcmp(name, "list_cap")) { ctx->opt_list_cap = value; return IPKGPU_OK; }
    if (!strcmp(name, "variant")) { ctx->opt_variant = value; return IPKGPU_OK; }
    return fail(ctx, IPKGPU_ERR_INVALID, "unknown option '%s'", name);
}

}  // extern "C"

// ---- launch plumbing --------------------------------------------------------------------------
namespace {

constexpr int TW = 64;   // windows per tile
constexpr int NW = 4;    // wavefronts per workgroup

template <int SIGMA, int K> constexpr int fast_cap()
{
    if (SIGMA == 4) return K <= 10 ? 256 : 512;
    return 512;
}

template <int SIGMA, int K>
int launch_score(ipkgpu_ctx* ctx, const ScoreParams& p)
{
    constexpr int CAP = fast_cap<SIGMA, K>();
    constexpr size_t lds = TileGeo<SIGMA, K, TW>::HEAD_BYTES + (size_t)NW * wave_scratch_entries<SIGMA, K, CAP>() * 8;
    static_assert(lds <= 160 * 1024, "fast-path LDS budget");
    auto kern = score_tiles_kernel<SIGMA, K, CAP, TW, NW>;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t blocks = (uint64_t)p.n_batch_mats * p.tiles_per_mat;
    if (blocks > 0x7fffffffull) return fail(ctx, IPKGPU_ERR_INVALID, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(NW * 64), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return IPKGPU_OK;
}

template <int SIGMA, int K>
int launch_overflow(ipkgpu_ctx* ctx, const ScoreParams& p)
{
    if constexpr (Geo<SIGMA, K, 1 << 30>::DIRECT || ipow(SIGMA, K - K / 2) <= (uint32_t)fast_cap<SIGMA, K>()) {
        (void)ctx; (void)p;
        return IPKGPU_OK;                           // lists can never overflow
    } else {
        constexpr size_t lds = TileGeo<SIGMA, K, 1>::HEAD_BYTES + (size_t)wave_scratch_entries<SIGMA, K, 1 << 30>() * 8;
        static_assert(lds <= 160 * 1024, "big-list LDS budget");
        auto kern = score_overflow_kernel<SIGMA, K>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
        hipLaunchKernelGGL(kern, dim3(ctx->num_cu * per_cu), dim3(64), lds, ctx->stream, p);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}

template <int SIGMA, int K>
int launch_both(ipkgpu_ctx* ctx, const ScoreParams& p)
{
    int rc = launch_score<SIGMA, K>(ctx, p);
    if (rc) return rc;
    return launch_overflow<SIGMA, K>(ctx, p);
}

int dispatch_score(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const ScoreParams& p)
{
    if (sigma == 4) {
        switch (k) {
            case 2: return launch_both<4, 2>(ctx, p);
            case 3: return launch_both<4, 3>(ctx, p);
            case 4: return launch_both<4, 4>(ctx, p);
            case 5: return launch_both<4, 5>(ctx, p);
            case 6: return launch_both<4, 6>(ctx, p);
            case 7: return launch_both<4, 7>(ctx, p);
            case 8: return launch_both<4, 8>(ctx, p);
            case 9: return launch_both<4, 9>(ctx, p);
            case 10: return launch_both<4, 10>(ctx, p);
            case 11: return launch_both<4, 11>(ctx, p);
            case 12: return launch_both<4, 12>(ctx, p);
        }
    } else if (sigma == 20) {
        switch (k) {
            case 2: return launch_both<20, 2>(ctx, p);
            case 3: return launch_both<20, 3>(ctx, p);
            case 4: return launch_both<20, 4>(ctx, p);
            case 5: return launch_both<20, 5>(ctx, p);
            case 6: return launch_both<20, 6>(ctx, p);
        }
    }
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k (%u/%u)", sigma, k);
}

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};

}  // namespace

extern "C" {

int ipkgpu_score_groups_device(ipkgpu_ctx* ctx, const float* logp_dev, uint32_t n_mats, uint32_t sites,
                               uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                               ipkgpu_result** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!logp_dev || !mat_group) return fail(ctx, IPKGPU_ERR_INVALID, "null input pointer");
    if (sigma != 4 && sigma != 20) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported alphabet size %u (4 or 20)", sigma);
    if (k < 2 || k > ipkgpu_max_k(sigma))
        return fail(ctx, IPKGPU_ERR_INVALID, "k=%u out of range [2, %u] for sigma=%u", k, ipkgpu_max_k(sigma), sigma);
    if (n_mats == 0) return fail(ctx, IPKGPU_ERR_INVALID, "no matrices");
    if (sites < k) return fail(ctx, IPKGPU_ERR_INVALID, "alignment has %u sites, fewer than k=%u", sites, k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // group discovery: first-seen order of branch ids (group_ghost_ids, db_builder.cpp:524-553)
    std::vector<uint32_t> group_ids;
    std::vector<uint32_t> slot_of(n_mats);
    {
        std::unordered_map<uint32_t, uint32_t> index;
        index.reserve(n_mats);
        for (uint32_t i = 0; i < n_mats; ++i) {
            auto it = index.find(mat_group[i]);
            if (it == index.end()) {
                it = index.emplace(mat_group[i], (uint32_t)group_ids.size()).first;
                group_ids.push_back(mat_group[i]);
            }
            slot_of[i] = it->second;
        }
    }
    const uint32_t n_groups = (uint32_t)group_ids.size();
    const uint64_t table_size = ipow(sigma, (int)k);
    const uint64_t table_bytes = table_size * 4;
    uint64_t gpb = std::max<uint64_t>(1, (uint64_t)ctx->workspace_bytes / table_bytes);
    gpb = std::min<uint64_t>(gpb, n_groups);
    const uint32_t chunks_per_group = (uint32_t)((table_size + CHUNK - 1) / CHUNK);
    while (gpb > 1 && gpb * chunks_per_group > 0x7fffffffull) gpb /= 2;
    const uint32_t nwin = sites - k + 1;
    const uint32_t tiles_per_mat = (nwin + TW - 1) / TW;

    ipkgpu_result* res = new (std::nothrow) ipkgpu_result();
    if (!res) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    res->ctx = ctx;
    res->group_ids = group_ids;
    res->offsets.assign((size_t)n_groups + 1, 0);
    auto bail = [&](int rc) { ipkgpu_result_free(res); return rc; };
#define TRY_RC(expr) do { int rc_ = (expr); if (rc_) return bail(rc_); } while (0)
#define HIP_TRY_R(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return bail(fail(ctx, e_ == hipErrorOutOfMemory ? IPKGPU_ERR_NOMEM : IPKGPU_ERR_HIP, \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)

    TRY_RC(ensure(ctx, &ctx->best, &ctx->best_cap, (size_t)n_mats * (sites + 1) * 4));
    TRY_RC(ensure(ctx, &ctx->table, &ctx->table_cap, (size_t)(gpb * table_bytes)));
    TRY_RC(ensure(ctx, &ctx->idx, &ctx->idx_cap, (size_t)n_mats * 8));
    TRY_RC(ensure(ctx, &ctx->counts, &ctx->counts_cap, (size_t)(gpb * chunks_per_group) * 4));
    TRY_RC(ensure(ctx, &ctx->offsets, &ctx->offsets_cap, (size_t)(gpb * chunks_per_group + 1) * 8));
    TRY_RC(ensure(ctx, &ctx->goff, &ctx->goff_cap, (size_t)(gpb + 1) * 8));

    std::vector<hipEvent_t> events;
    auto new_event = [&]() { hipEvent_t e = nullptr; hipEventCreate(&e); events.push_back(e); return e; };
    auto record = [&](hipEvent_t e) { return hipEventRecord(e, ctx->stream); };
    hipEvent_t ev_begin = new_event(), ev_pre = new_event(), ev_end = new_event();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_score, ev_compact;

    unsigned long long* d_emitted = reinterpret_cast<unsigned long long*>(ctx->small);
    uint32_t* d_ovf_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 16);
    HIP_TRY_R(hipMemsetAsync(ctx->small, 0, 64, ctx->stream));

    HIP_TRY_R(record(ev_begin));
    if (sigma == 4)
        hipLaunchKernelGGL(prefix_max_kernel<4>, dim3(n_mats), dim3(256), 0, ctx->stream, logp_dev, sites, (float*)ctx->best);
    else
        hipLaunchKernelGGL(prefix_max_kernel<20>, dim3(n_mats), dim3(256), 0, ctx->stream, logp_dev, sites, (float*)ctx->best);
    HIP_TRY_R(hipGetLastError());
    HIP_TRY_R(record(ev_pre));

    // batches of groups whose tables fit the workspace
    std::vector<uint32_t> idx_host((size_t)n_mats * 2);
    uint64_t total_entries = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += (uint32_t)gpb) {
        const uint32_t gb = std::min<uint32_t>((uint32_t)gpb, n_groups - g0);
        uint32_t nb = 0;
        uint32_t* mat_list_h = idx_host.data();
        uint32_t* mat_slot_h = idx_host.data() + n_mats;
        for (uint32_t i = 0; i < n_mats; ++i) {
            mat_slot_h[i] = 0;
            if (slot_of[i] >= g0 && slot_of[i] < g0 + gb) { mat_list_h[nb++] = i; mat_slot_h[i] = slot_of[i] - g0; }
        }
        HIP_TRY_R(hipMemcpyAsync(ctx->idx, idx_host.data(), (size_t)n_mats * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY_R(hipStreamSynchronize(ctx->stream));     // idx_host is reused by the next batch
        TRY_RC(ensure(ctx, &ctx->ovfq, &ctx->ovfq_cap, (size_t)nb * nwin * 8));

        ScoreParams p;
        p.logp = logp_dev;
        p.best = (const float*)ctx->best;
        p.mat_list = (const uint32_t*)ctx->idx;
        p.mat_slot = (const uint32_t*)ctx->idx + n_mats;
        p.n_batch_mats = nb; p.sites = sites; p.nwin = nwin; p.tiles_per_mat = tiles_per_mat;
        p.eps = log_eps;
        p.table = (uint32_t*)ctx->table;
        p.table_size = table_size;
        p.emitted = d_emitted;
        p.ovf_queue = (unsigned long long*)ctx->ovfq;
        p.ovf_count = d_ovf_count;
        p.flags = (uint32_t)(ctx->opt_variant == 99 ? 1 : 0);

        hipEvent_t s0 = new_event(), s1 = new_event(), c1 = new_event();
        HIP_TRY_R(hipMemsetAsync(ctx->table, 0, (size_t)gb * table_bytes, ctx->stream));
        HIP_TRY_R(hipMemsetAsync(d_ovf_count, 0, 4, ctx->stream));
        HIP_TRY_R(record(s0));
        TRY_RC(dispatch_score(ctx, sigma, k, p));
        HIP_TRY_R(record(s1));
        ev_score.push_back({s0, s1});
        res->score_launches += 1;

        const uint32_t n_chunks = gb * chunks_per_group;
        hipLaunchKernelGGL(count_chunks_kernel, dim3(n_chunks), dim3(256), 0, ctx->stream,
                           (const uint32_t*)ctx->table, table_size, chunks_per_group, (uint32_t*)ctx->counts);
        HIP_TRY_R(hipGetLastError());
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           (const uint32_t*)ctx->counts, (uint64_t)n_chunks, total_entries, (uint64_t*)ctx->offsets);
        HIP_TRY_R(hipGetLastError());
        // group offsets of this batch (every chunks_per_group-th offset) + new total
        std::vector<uint64_t> goff((size_t)gb + 1);
        hipLaunchKernelGGL(gather_offsets_kernel, dim3((gb + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                           (const uint64_t*)ctx->offsets, chunks_per_group, gb + 1, (uint64_t*)ctx->goff);
        HIP_TRY_R(hipGetLastError());
        HIP_TRY_R(hipMemcpyAsync(goff.data(), ctx->goff, ((size_t)gb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY_R(hipStreamSynchronize(ctx->stream));
        for (uint32_t g = 0; g <= gb; ++g) res->offsets[g0 + g] = goff[g];
        const uint64_t new_total = goff[gb];
        if (new_total > res->cap) {
            // grow the output (exact for a single batch; doubling across batches)
            const size_t new_cap = (g0 + gb >= n_groups) ? (size_t)new_total : (size_t)std::max<uint64_t>(new_total, 2 * res->cap);
            uint32_t* nk = nullptr; float* ns = nullptr;
            HIP_TRY_R(hipMalloc((void**)&nk, std::max<size_t>(new_cap, 1) * 4));
            hipError_t e2 = hipMalloc((void**)&ns, std::max<size_t>(new_cap, 1) * 4);
            if (e2 != hipSuccess) { hipFree(nk); HIP_TRY_R(e2); }
            if (total_entries) {
                hipMemcpyAsync(nk, res->d_keys, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                hipMemcpyAsync(ns, res->d_scores, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                hipStreamSynchronize(ctx->stream);
            }
            if (res->d_keys) hipFree(res->d_keys);
            if (res->d_scores) hipFree(res->d_scores);
            res->d_keys = nk; res->d_scores = ns; res->cap = new_cap;
        }
        if (sigma == 4)
            hipLaunchKernelGGL(write_chunks_kernel<4>, dim3(n_chunks), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->table,
                               table_size, chunks_per_group, (int)k, (const uint64_t*)ctx->offsets, res->d_keys, res->d_scores);
        else
            hipLaunchKernelGGL(write_chunks_kernel<20>, dim3(n_chunks), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->table,
                               table_size, chunks_per_group, (int)k, (const uint64_t*)ctx->offsets, res->d_keys, res->d_scores);
        HIP_TRY_R(hipGetLastError());
        HIP_TRY_R(record(c1));
        ev_compact.push_back({s1, c1});
        total_entries = new_total;
    }
    HIP_TRY_R(record(ev_end));
    unsigned long long emitted = 0;
    HIP_TRY_R(hipMemcpyAsync(&emitted, d_emitted, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY_R(hipStreamSynchronize(ctx->stream));
    res->emitted = emitted;

    float ms = 0;
    hipEventElapsedTime(&ms, ev_begin, ev_end); res->t_total = ms;
    hipEventElapsedTime(&ms, ev_begin, ev_pre); res->t_prefix = ms;
    for (auto& pr : ev_score) { hipEventElapsedTime(&ms, pr.first, pr.second); res->t_score += ms; }
    for (auto& pr : ev_compact) { hipEventElapsedTime(&ms, pr.first, pr.second); res->t_compact += ms; }
    for (hipEvent_t e : events) hipEventDestroy(e);
#undef TRY_RC
#undef HIP_TRY_R
    *out = res;
    return IPKGPU_OK;
}

int ipkgpu_score_groups(ipkgpu_ctx* ctx, const float* logp, uint32_t n_mats, uint32_t sites, uint32_t sigma,
                        const uint32_t* mat_group, uint32_t k, float log_eps, ipkgpu_result** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!logp || !mat_group) return fail(ctx, IPKGPU_ERR_INVALID, "null input pointer");
    if (sigma != 4 && sigma != 20) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported alphabet size %u (4 or 20)", sigma);
    if (n_mats == 0) return fail(ctx, IPKGPU_ERR_INVALID, "no matrices");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_mats * sites * sigma * 4;
    float* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, std::max<size_t>(bytes, 4)));
    hipError_t e = hipMemcpy(d, logp, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(d); HIP_TRY(ctx, e); }
    const int rc = ipkgpu_score_groups_device(ctx, d, n_mats, sites, sigma, mat_group, k, log_eps, out);
    hipFree(d);
    return rc;
}

uint32_t ipkgpu_result_num_groups(const ipkgpu_result* r) { return r ? (uint32_t)r->group_ids.size() : 0; }
const uint32_t* ipkgpu_result_group_ids(const ipkgpu_result* r) { return r ? r->group_ids.data() : nullptr; }
const uint64_t* ipkgpu_result_offsets(const ipkgpu_result* r) { return r ? r->offsets.data() : nullptr; }
uint64_t ipkgpu_result_emitted(const ipkgpu_result* r) { return r ? r->emitted : 0; }
const uint32_t* ipkgpu_result_keys_device(const ipkgpu_result* r) { return r ? r->d_keys : nullptr; }
const float* ipkgpu_result_scores_device(const ipkgpu_result* r) { return r ? r->d_scores : nullptr; }

const uint32_t* ipkgpu_result_keys(ipkgpu_result* r)
{
    if (!r) return nullptr;
    if (!r->h_keys_ok) {
        const size_t n = (size_t)r->offsets.back();
        r->h_keys.resize(std::max<size_t>(n, 1));
        hipSetDevice(r->ctx->device);
        if (n && hipMemcpy(r->h_keys.data(), r->d_keys, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        r->h_keys_ok = true;
    }
    return r->h_keys.data();
}

const float* ipkgpu_result_scores(ipkgpu_result* r)
{
    if (!r) return nullptr;
    if (!r->h_scores_ok) {
        const size_t n = (size_t)r->offsets.back();
        r->h_scores.resize(std::max<size_t>(n, 1));
        hipSetDevice(r->ctx->device);
        if (n && hipMemcpy(r->h_scores.data(), r->d_scores, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        r->h_scores_ok = true;
    }
    return r->h_scores.data();
}

double ipkgpu_result_time_ms(const ipkgpu_result* r, int which)
{
    if (!r) return 0;
    switch (which) {
        case IPKGPU_T_TOTAL: return r->t_total;
        case IPKGPU_T_PREFIX: return r->t_prefix;
        case IPKGPU_T_SCORE: return r->t_score;
        case IPKGPU_T_COMPACT: return r->t_compact;
        case IPKGPU_T_SCORE_LAUNCHES: return (double)r->score_launches;
    }
    return 0;
}

void ipkgpu_result_free(ipkgpu_result* r)
{
    if (!r) return;
    if (r->ctx) hipSetDevice(r->ctx->device);
    if (r->d_keys) hipFree(r->d_keys);
    if (r->d_scores) hipFree(r->d_scores);
    delete r;
}

}  // extern "C"
