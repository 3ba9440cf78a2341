// ipkgpu.hip -- MI355X (gfx950) phylo-k-mer scoring engine: kernels + C ABI (include/ipkgpu.h).
//
// Path accelerated (IPK tree): db_builder.cpp:576-698 (explore_kmers / explore_group) ->
// window.cpp:16-27,164-182 (prefix of column maxima, sliding windows) -> pk_compute.cpp:42-114
// (DCLA::DC) -> branch_group.cpp:88-101 (put: per-branch max-reduce).
//
// Data layout in HBM:
//   logp   [n_mats][sites][sigma] f32        caller's matrices, read once per scoring pass
//   best   [n_mats][sites+1]      f32        sequential float prefix sums of column maxima
//   pool   [chunks][256] {u32 code, f32 score}   every scored phylo-k-mer once (stream variant), chunked
//                                            per (wavefront, key bucket); desc[chunk] = (group, bucket, count)
//   table  [groups_in_batch][sigma^k] u32    per-group dense max table, order-preserving score
//                                            codes, 0 = empty (the on-device group_hash_map)
//   results: group-major CSR (offsets, keys, scores) or key-major parts / database shards
//            (counts per key, {branch, score} entries) -- see include/ipkgpu.h
//
// There is NO CPU fallback: without a GPU ipkgpu_create fails with IPKGPU_ERR_NODEVICE.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <memory>
#include <vector>

#include "../../include/ipkgpu.h"
#include "dcla_device.hpp"
#include "kernels_score.hpp"
#include "kernels_quad.hpp"
#include "kernels_compact.hpp"
#include "kernels_keymajor.hpp"
#include "kernels_reduce_pipe.hpp"
#include "kernels_filter.hpp"
#include "kernels_dbfile.hpp"
#include <chrono>

using namespace ipkgpu;

// =============================================================================================
// host side
// =============================================================================================

namespace {

struct DevBuf {                       // grow-only device workspace
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// events on a stream, read after a wait
// (events are recycled through a small per-context pool: creating and destroying a dozen of them cost 20 us of every call)
struct EventPool {
    std::vector<hipEvent_t> free_ev;
    hipEvent_t get()
    {
        if (!free_ev.empty()) { hipEvent_t e = free_ev.back(); free_ev.pop_back(); return e; }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
    void put(hipEvent_t e) { if (free_ev.size() < 64) free_ev.push_back(e); else (void)hipEventDestroy(e); }
};
struct Stopwatch {
    hipStream_t stream;
    EventPool* pool;                  // the context's (events belong to its device)
    std::vector<hipEvent_t> ev;
    Stopwatch(hipStream_t s, EventPool* p) : stream(s), pool(p) {}
    ~Stopwatch() { for (hipEvent_t e : ev) pool->put(e); }
    int mark()
    {
        hipEvent_t e = pool->get();
        if (!e) return -1;
        (void)hipEventRecord(e, stream);
        ev.push_back(e);
        return (int)ev.size() - 1;
    }
    double ms(int a, int b) const
    {
        float t = 0;
        if (a < 0 || b < 0 || hipEventElapsedTime(&t, ev[a], ev[b]) != hipSuccess) return 0;
        return t;
    }
};

}  // namespace

struct ipkgpu_comm;
struct ipkgpu_ctx;
static void ipkgpu_comm_release(ipkgpu_ctx* ctx);
struct ipkgpu_ctx {
    int device = 0;
    ipkgpu_comm* comm = nullptr;      // RCCL communicator of the k-mer-keyed exchange (comm_rccl.hpp), if initialised
    hipStream_t stream = nullptr;
    std::string err;
    int64_t workspace_bytes = 0;
    int64_t opt_variant = 0;
    int64_t opt_flags = 0;
    int64_t opt_kmc_pass = 0;            // groups per pass of the compressed key-major writer (0: IPK_KMC_PASS)
    int64_t opt_prefix_mats = 0;         // matrices per workgroup of prefix_max_kernel (0: by the matrix count; 1, 2, 4, 8: tests)
    int64_t opt_wg_chunks2 = 0;       // tuning knob: overrides IPK_WG_CHUNKS2 (0 = built-in), opt_rounds: IPK_ROUNDS
    int64_t opt_rounds = 0;
    int64_t opt_pool_limit = 0;       // test knob: bytes the pair pool may take (0 = what the device has free)
    int64_t opt_pool_chunks = 0;      // test knob: size of the FIRST pair-pool attempt (forces the grow-and-redo path)
    DevBuf table, best, ovfq, counts, offsets, goff, idx, branch, scan_sums, scan_boff, tmp_a, tmp_b, tmp_c;
    DevBuf pool, desc, gbcnt, gboff, gbcur, clist, gm, tile_next;   // stream variant: pair pool, chunk descriptors, chunk index, tile counters
    DevBuf ptrs;                 // per-source pointer arrays of a merge
    DevBuf mask;                 // occupancy bits of ctx->table ([groups in batch][mask_words]) when mask_valid
    bool mask_valid = false;
    uint64_t mask_words = 0;     // 2 * ceil(table_size / 64): rows padded to whole 64-slot blocks
    // compressed table form (exact-partition variant on sparse key spaces; comp_table.hpp): no dense ctx->table
    bool table_compressed = false;
    DevBuf rank, vaddr, ucnt, qpack, xstart, pcounts;
    DevBuf cvals, coff, croom;          // compressed output of the chunk-fed reduce: values, their offsets per (group, bucket), room
    bool comp_own_vals = false;         // the compressed values live in cvals / coff (chunked pool) instead of in place in the pool
    std::vector<uint32_t> h_branch;     // host copy of the last call's branch ids (source of an asynchronous upload)
    uint32_t comp_nb = 0, comp_stride = 0, comp_tbl = 0;
    double pairs_per_window = 0;      // calibration of the pair pool from the previous call
    double acc_main_ms = 0, acc_reduce_ms = 0;   // dominant scoring kernel / LDS reduce pass of the current call
    double acc_count_ms = 0, acc_write_ms = 0, acc_km_ms = 0;   // exact-partition count / write pass, key-major writer
    const char* main_kernel = "";                // name of the dominant kernel of the last scoring call
    // caching allocator for result buffers: hipMalloc/hipFree of multi-GB blocks costs 10-100 ms, so
    // released result buffers are kept (bounded) and handed out again to the next call
    std::vector<std::pair<void*, size_t>> free_blocks;
    std::unordered_map<void*, size_t> live_blocks;
    size_t cached_bytes = 0, cache_limit = 0;
    void* small = nullptr;            // emitted (u64) @0, ovf_count (u32) @16
    // pinned staging of the small per-batch index uploads (matrix lists, group -> matrices CSR): copied from without a wait;
    // up_done is recorded behind the last copy and waited for only before the staging is written again
    void* h_up = nullptr; size_t h_up_cap = 0; hipEvent_t up_done = nullptr; bool up_pending = false;
    unsigned long long emitted_host = 0; bool emitted_fetched = false;   // the batch's scored-k-mer count, read back with the batch's last wait
    // Read-backs without a wait of their own: pinned words the stream copies into, valid after the NEXT wait on the stream.
    // h_rb[0..15] = ctx->small (scored k-mers @0, big-list queue length @4, chunk ids drawn @8, pool-exhausted flag @9);
    // h_rb64[RB_OWNER_OFF ..] = a key-major batch's owner offsets (RB_OWNERS_MAX + 1 of them at most).
    uint32_t* h_rb = nullptr;
    EventPool events;
    // A stream-variant batch whose pool check and statistics are still owed (score_batch_finish): the call did not wait after
    // pass 1 (spec_skip_wait: the previous call had few big-list windows -- those then take the atomic kernel whatever their number,
    // which is always correct on dense tables) and, on request, not at its end either.
    bool spec_skip_wait = false;
    uint64_t pool_want_min = 0;         // chunks the pool must hold on the redo of a batch whose pool ran out
    struct Pending {
        bool active = false;
        std::unique_ptr<Stopwatch> sw; int ev_a = -1, ev_b = -1, ev_c = -1, ev_d = -1;
        uint64_t cap = 0, max_chunks = 0, windows = 0; uint32_t gb = 0; bool pool_ovf_ok = false, use_quad = false;
    } pend;
    uint64_t last_entries = 0; uint32_t last_gb = 0;   // the key-major writer's output of the previous batch: the next one's estimate
    double t_write_total = 0, t_write_device = 0, t_write_file = 0;   // last ipkgpu_db_write
    int num_cu = 256;
};

struct ipkgpu_result {
    ipkgpu_ctx* ctx = nullptr;
    std::vector<uint32_t> group_ids;
    std::vector<uint64_t> offsets;
    uint64_t emitted = 0;
    uint32_t* d_keys = nullptr;
    float* d_scores = nullptr;
    uint32_t* d_positions = nullptr;          // KEEP_POSITIONS variant only
    size_t cap = 0;
    std::vector<uint32_t> h_keys;
    std::vector<float> h_scores;
    std::vector<uint32_t> h_positions;
    bool h_keys_ok = false, h_scores_ok = false, h_positions_ok = false;
    double t_total = 0, t_prefix = 0, t_score = 0, t_compact = 0, t_main = 0, t_reduce = 0;
    int score_launches = 0;
};

struct ipkgpu_parts {
    ipkgpu_ctx* ctx = nullptr;
    uint32_t n_owners = 1;
    uint64_t slots = 0;                       // padded key slots per owner = ceil(sigma^k / n_owners)
    uint32_t* d_counts = nullptr;             // [n_owners][slots]
    uint2* d_entries = nullptr;               // owner-major, key-major, group order: (branch, score bits)
    std::vector<uint64_t> owner_off;          // [n_owners + 1] entry offsets
    uint64_t emitted = 0;
    // one owner, one batch: the database's key list (the non-empty slots and their entry offsets) is built inside the scoring
    // call, ahead of the key-major writer -- ipkgpu_db_from_parts then only hands the arrays over
    uint32_t* pre_keys = nullptr; uint64_t* pre_key_off = nullptr; uint64_t pre_n_keys = 0; double t_keys = 0;
    double t_total = 0, t_prefix = 0, t_score = 0, t_compact = 0, t_main = 0, t_reduce = 0, t_count = 0, t_write = 0, t_km = 0;
    int score_launches = 0;
};

struct ipkgpu_db {
    ipkgpu_ctx* ctx = nullptr;
    uint64_t n_keys = 0, n_entries = 0;
    uint32_t* d_keys = nullptr;               // [n_keys] packed codes, ascending
    uint64_t* d_key_off = nullptr;            // [n_keys + 1]
    uint2* d_entries = nullptr;               // [n_entries] (branch, score bits)
    std::vector<uint32_t> h_keys;
    std::vector<uint64_t> h_key_off;
    std::vector<uint32_t> h_entries;          // [n_entries][2]
    bool h_ok = false;
    double t_merge = 0;
    // filter stage (row n1)
    double* d_fv64 = nullptr;
    float* d_fv32 = nullptr;
    uint32_t* d_order = nullptr;              // positions of the k-mers sorted by (filter value, key)
    std::vector<double> h_fv64;
    std::vector<float> h_fv32;
    std::vector<uint32_t> h_order;
    bool h_filter_ok = false;
    double t_filter = 0;
};

static std::string g_create_err;
constexpr size_t RB_BYTES = 4096, RB_OWNER_OFF = 8 /* in 64-bit words */, RB_OWNERS_MAX = 500;

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
#define RC_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// A workspace of at least `need` bytes.  One that has to GROW gets a sixteenth more than asked for: several sizes follow counts that vary a little
// from call to call (chunks drawn, values per slice), and a buffer grown to the byte paid hipFree + hipMalloc of the whole block whenever the next
// call needed one chunk more -- 1.2 s of a 0.1-s step for the 30-GB buffers of all of cfg3 on one GPU (tools/step_trace.py cfg3 1000).
static int ensure(ipkgpu_ctx* ctx, DevBuf& b, size_t need)
{
    if (b.cap >= need && b.p) return IPKGPU_OK;
    const bool regrow = b.p != nullptr;
    if (b.p) { HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    need = std::max<size_t>(need, 16);
    size_t want = regrow ? need + need / 16 : need;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess && want > need) { (void)hipGetLastError(); want = need; e = hipMalloc(&b.p, want); }   // (no room for the margin)
    HIP_TRY(ctx, e);
    b.cap = want;
    return IPKGPU_OK;
}


// ---- caching device allocator for result buffers ---------------------------------------------------
// Pinned upload staging of `bytes`: waits for the copies still reading the previous contents, grows if needed.
static int upload_stage(ipkgpu_ctx* ctx, size_t bytes, void** out)
{
    if (ctx->up_pending) { HIP_TRY(ctx, hipEventSynchronize(ctx->up_done)); ctx->up_pending = false; }
    if (!ctx->up_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_done, hipEventDisableTiming));
    if (bytes > ctx->h_up_cap) {
        if (ctx->h_up) { (void)hipHostFree(ctx->h_up); ctx->h_up = nullptr; ctx->h_up_cap = 0; }
        const size_t cap = std::max<size_t>(bytes + bytes / 2, 1 << 16);
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_up, cap, hipHostMallocDefault));
        ctx->h_up_cap = cap;
    }
    *out = ctx->h_up;
    return IPKGPU_OK;
}
static int upload_staged(ipkgpu_ctx* ctx)       // behind the last copy out of the staging
{
    HIP_TRY(ctx, hipEventRecord(ctx->up_done, ctx->stream));
    ctx->up_pending = true;
    return IPKGPU_OK;
}

static hipError_t ctx_alloc(ipkgpu_ctx* ctx, void** out, size_t bytes)
{
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->free_blocks.size(); ++i) {
        const size_t cap = ctx->free_blocks[i].second;
        if (cap >= bytes && cap <= bytes + bytes / 2 + (1 << 20) && (best == (size_t)-1 || cap < ctx->free_blocks[best].second)) best = i;
    }
    if (best != (size_t)-1) {
        *out = ctx->free_blocks[best].first;
        ctx->live_blocks[*out] = ctx->free_blocks[best].second;
        ctx->cached_bytes -= ctx->free_blocks[best].second;
        ctx->free_blocks.erase(ctx->free_blocks.begin() + best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && !ctx->free_blocks.empty()) {          // out of memory: drop the cache and retry
        for (auto& b : ctx->free_blocks) (void)hipFree(b.first);
        ctx->free_blocks.clear(); ctx->cached_bytes = 0;
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) ctx->live_blocks[*out] = bytes;
    return e;
}
static void ctx_release(ipkgpu_ctx* ctx, void* p)
{
    if (!p) return;
    auto it = ctx->live_blocks.find(p);
    if (it == ctx->live_blocks.end()) { (void)hipFree(p); return; }
    const size_t cap = it->second;
    ctx->live_blocks.erase(it);
    if (ctx->cached_bytes + cap <= ctx->cache_limit) { ctx->free_blocks.push_back({p, cap}); ctx->cached_bytes += cap; }
    else (void)hipFree(p);
}

// Kernels that give a slot (or a key) a whole wavefront are launched over at most this many items at a time: a HIP launch takes fewer than
// 2^32 threads, and 4^13 slots x 64 lanes is exactly that.
static constexpr uint64_t WAVE_PER_ITEM_SPAN = 1ull << 24;

// A block of at least min_bytes, preferably want_bytes: a cached block that holds min_bytes is taken whole (*got = its size) --
// the result buffer of the previous, equally shaped call serves the next one without a new hipMalloc.
static hipError_t ctx_alloc_atleast(ipkgpu_ctx* ctx, void** out, size_t min_bytes, size_t want_bytes, size_t* got)
{
    min_bytes = std::max<size_t>((min_bytes + 255) & ~(size_t)255, 256);
    want_bytes = std::max(want_bytes, min_bytes);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->free_blocks.size(); ++i) {
        const size_t cap = ctx->free_blocks[i].second;
        if (cap >= min_bytes && cap <= want_bytes + want_bytes / 2 + (1 << 20) && (best == (size_t)-1 || cap < ctx->free_blocks[best].second)) best = i;
    }
    if (best != (size_t)-1) {
        *out = ctx->free_blocks[best].first;
        *got = ctx->free_blocks[best].second;
        ctx->live_blocks[*out] = *got;
        ctx->cached_bytes -= *got;
        ctx->free_blocks.erase(ctx->free_blocks.begin() + best);
        return hipSuccess;
    }
    const hipError_t e = ctx_alloc(ctx, out, want_bytes);
    if (e == hipSuccess) *got = ctx->live_blocks[*out];
    return e;
}

extern "C" {

uint32_t ipkgpu_bits_per_symbol(uint32_t sigma) { return sigma == 4 ? 2u : sigma == 20 ? 5u : 0u; }
uint32_t ipkgpu_max_k(uint32_t sigma) { return sigma == 4 ? 14u : sigma == 20 ? 6u : 0u; }
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
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { ctx->workspace_bytes = (int64_t)(free_b / 2); ctx->cache_limit = free_b / 4; }
    else ctx->workspace_bytes = (int64_t)8 << 30;
    if ((e = hipMalloc(&ctx->small, 64)) != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(nullptr, IPKGPU_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    if ((e = hipHostMalloc((void**)&ctx->h_rb, RB_BYTES, hipHostMallocDefault)) != hipSuccess) {
        (void)hipFree(ctx->small);
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(nullptr, IPKGPU_ERR_NOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    memset(ctx->h_rb, 0, RB_BYTES);
    *out = ctx;
    return IPKGPU_OK;
}

void ipkgpu_destroy(ipkgpu_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->table, &ctx->best, &ctx->ovfq, &ctx->counts, &ctx->offsets, &ctx->goff, &ctx->idx,
                      &ctx->branch, &ctx->scan_sums, &ctx->scan_boff, &ctx->tmp_a, &ctx->tmp_b, &ctx->tmp_c,
                      &ctx->pool, &ctx->desc, &ctx->gbcnt, &ctx->gboff, &ctx->gbcur, &ctx->clist, &ctx->gm, &ctx->tile_next, &ctx->mask,
                      &ctx->rank, &ctx->vaddr, &ctx->ucnt, &ctx->qpack, &ctx->xstart, &ctx->pcounts, &ctx->ptrs,
                      &ctx->cvals, &ctx->coff, &ctx->croom};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto& b : ctx->free_blocks) (void)hipFree(b.first);
    ipkgpu_comm_release(ctx);
    if (ctx->small) (void)hipFree(ctx->small);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    ctx->pend.sw.reset();
    for (hipEvent_t e : ctx->events.free_ev) (void)hipEventDestroy(e);
    ctx->events.free_ev.clear();
    if (ctx->h_rb) (void)hipHostFree(ctx->h_rb);
    if (ctx->up_done) (void)hipEventDestroy(ctx->up_done);
    (void)hipStreamDestroy(ctx->stream);
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
    if (!strcmp(name, "variant")) { ctx->opt_variant = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_flags")) { ctx->opt_flags = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_pool_chunks")) { ctx->opt_pool_chunks = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_pool_limit_bytes")) { ctx->opt_pool_limit = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_wg_chunks2")) { ctx->opt_wg_chunks2 = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_rounds")) { ctx->opt_rounds = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_kmc_pass")) { ctx->opt_kmc_pass = value; return IPKGPU_OK; }
    if (!strcmp(name, "debug_prefix_mats")) { ctx->opt_prefix_mats = value; return IPKGPU_OK; }
    return fail(ctx, IPKGPU_ERR_INVALID, "unknown option '%s'", name);
}

}  // extern "C"

// ---- launch plumbing --------------------------------------------------------------------------
namespace {

constexpr int TW = 128;  // windows per tile
constexpr int NW = 8;    // wavefronts per workgroup

#ifndef IPK_QCAP
#define IPK_QCAP 160
#endif
#ifndef IPK_AACAP
#define IPK_AACAP 512
#endif
// DNA k = 11, 12 (row-per-lane join, one window's child nodes at a time): a wavefront needs 2 KB of child lists, 2 x 384 half-list
// entries and 4 KB of bucket words = 12 KB; three workgroups of FOUR wavefronts with 32-window tiles fit a CU = 12 wavefronts
// (round 2: capacity 416, 64-window tiles, three workgroups of three = 9).  cfg3 share, scoring kernel: 6.15 ms at 9 wavefronts,
// 5.58 ms at 12; capacity 352 / 320 trade the same occupancy for more big-list windows (step 15.0 / 15.1 ms against 14.6).
#ifndef IPK_QCAP12
#define IPK_QCAP12 384
#endif
#ifndef IPK_QNW12
#define IPK_QNW12 4
#endif
#ifndef IPK_QTW12
#define IPK_QTW12 32
#endif
template <int SIGMA, int K> constexpr int fast_cap()
{
    if (SIGMA == 4) return K <= 10 ? IPK_QCAP : IPK_QCAP12;
    return IPK_AACAP;
}

template <int SIGMA, int K, bool POS = false>
int launch_score(ipkgpu_ctx* ctx, const ScoreParams& p)
{
    constexpr int CAP = fast_cap<SIGMA, K>();
    constexpr size_t lds = TileGeo<SIGMA, K, TW>::HEAD_BYTES + (size_t)NW * wave_scratch_entries<SIGMA, K, CAP>() * 8;
    static_assert(lds <= 160 * 1024, "fast-path LDS budget");
    auto kern = score_tiles_kernel<SIGMA, K, CAP, TW, NW, POS>;
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t blocks = (uint64_t)p.n_batch_mats * p.tiles_per_mat;
    if (blocks > 0x7fffffffull) return fail(ctx, IPKGPU_ERR_INVALID, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(NW * 64), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return IPKGPU_OK;
}

template <int SIGMA, int K, bool POS = false>
int launch_overflow(ipkgpu_ctx* ctx, const ScoreParams& p)
{
    if constexpr (Geo<SIGMA, K, 1 << 30>::DIRECT || ipow(SIGMA, K - K / 2) <= (uint32_t)fast_cap<SIGMA, K>()) {
        (void)ctx; (void)p;
        return IPKGPU_OK;                           // lists can never overflow
    } else {
        constexpr size_t lds = TileGeo<SIGMA, K, 1>::HEAD_BYTES + (size_t)wave_scratch_entries<SIGMA, K, big_capf<SIGMA, K>()>() * 8;
        static_assert(lds <= 160 * 1024, "big-list LDS budget");
        auto kern = score_overflow_kernel<SIGMA, K, POS>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(32 / OVF_NW, (160 * 1024) / (lds + 64)));
        hipLaunchKernelGGL(kern, dim3(ctx->num_cu * per_cu), dim3(OVF_NW * 64), lds, ctx->stream, p);
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


// ---- stream variant (two-pass radix max-reduce) ----------------------------------------------------
template <int SIGMA, int K> constexpr uint32_t stream_tbl()
{
    constexpr uint64_t T = ipow(SIGMA, K);
    if (Geo<SIGMA, K, 1 << 30>::DIRECT) return 0;                       // sigma^k <= 64: nothing to gain
    if (SIGMA == 4 && K >= 13) return 0;                                // (a wavefront's 2048+ open chunks do not fit LDS: the exact partition takes over, xp_tbl)
    if (SIGMA == 4) return T <= 16384 ? (uint32_t)T : (K <= 10 ? 16384u : 32768u);
    if (K <= 3) return (uint32_t)T;                                     // 400, 8000
    if (K <= 5) return 32000u;                                          // 20^4 = 5 x 32000, 20^5 = 100 x 32000
    return 0;   // AA k=6: 2000 buckets per group = 16 KB of open-chunk state per WAVE; measured with two waves per workgroup:
                // pass 1 143 ms (occupancy-starved).  That key space takes the exact-partition variant instead (xp_tbl).
}

template <int SIGMA, int K> constexpr int stream_nw() { return NW; }
template <int SIGMA, int K> constexpr int stream_tw() { return TW; }

template <int SIGMA, int K>
int launch_stream_pass1(ipkgpu_ctx* ctx, const StreamParams& sp, uint32_t n_wg)
{
    constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
    if constexpr (TBL == 0) { (void)sp; (void)n_wg; return fail(ctx, IPKGPU_ERR_INVALID, "stream variant unsupported for this sigma/k"); }
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t T = ipow(SIGMA, K);
        constexpr uint32_t NB = (T + TBL - 1) / TBL;
        constexpr int SNW = stream_nw<SIGMA, K>(), STW = stream_tw<SIGMA, K>();
        constexpr size_t lds = TileGeo<SIGMA, K, STW>::HEAD_BYTES + (size_t)SNW * stream_wave_scratch<SIGMA, K, CAP>() * 8 + (size_t)SNW * 2 * NB * SUB * 4;
        static_assert(lds <= 160 * 1024, "stream pass-1 LDS budget");
        auto kern = score_stream_kernel<SIGMA, K, CAP, STW, SNW, TBL>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(SNW * 64), lds, ctx->stream, sp);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}

template <int SIGMA, int K>
int launch_stream_overflow(ipkgpu_ctx* ctx, const StreamParams& sp)
{
    constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
    if constexpr (TBL == 0 || ipow(SIGMA, K - K / 2) <= (uint32_t)fast_cap<SIGMA, K>()) {
        (void)ctx; (void)sp;
        return IPKGPU_OK;                           // lists can never overflow
    } else {
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        constexpr size_t lds = TileGeo<SIGMA, K, 1>::HEAD_BYTES + (size_t)wave_scratch_entries<SIGMA, K, big_capf<SIGMA, K>()>() * 8 +
                               (size_t)OVF_NW * 2 * NB * SUB * 4;
        static_assert(lds + 64 <= 160 * 1024, "big-list (stream) LDS budget");
        auto kern = score_overflow_stream_kernel<SIGMA, K, TBL>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(32 / OVF_NW, (160 * 1024) / (lds + 64)));
        hipLaunchKernelGGL(kern, dim3(ctx->num_cu * per_cu), dim3(OVF_NW * 64), lds, ctx->stream, sp);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}

#ifndef IPK_RPIPE_DEFAULT
#define IPK_RPIPE_DEFAULT 1
#endif
#ifndef IPK_RPIPE_D
#define IPK_RPIPE_D 2            // chunks per trip of the persistent reduce (two trips of D * 4 eight-byte loads per lane in flight)
#endif
template <int SIGMA, int K>
int launch_stream_pass2(ipkgpu_ctx* ctx, uint32_t n_gb, uint64_t T, uint32_t* table, bool compress)
{
    constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
    if constexpr (TBL == 0) { (void)n_gb; (void)T; (void)table; (void)compress; return fail(ctx, IPKGPU_ERR_INVALID, "stream variant unsupported"); }
    else {
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        constexpr int NT = TBL <= 16384 ? 512 : 1024;
        const size_t lds = compress ? (size_t)comp_padded_slots<TBL, NT>() * 4 + (NT / 64 + 1) * 4 : (size_t)TBL * 4;
        auto launch = [&](auto kern) -> int {
            if (lds > 64 * 1024)
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(n_gb), dim3(NT), lds, ctx->stream, ctx->pool.as<uint2>(),
                               ctx->gboff.as<uint64_t>(), ctx->clist.as<uint2>(), NB, T, table, ctx->mask.as<uint32_t>(), ctx->mask_words,
                               ctx->cvals.as<uint2>(), ctx->coff.as<uint64_t>(), ctx->rank.as<uint32_t>(), ctx->vaddr.as<uint64_t>(), ctx->ucnt.as<uint32_t>());
            HIP_TRY(ctx, hipGetLastError());
            return IPKGPU_OK;
        };
        // 128-KB slices (DNA k = 11, 12) leave one workgroup per CU: the persistent, pipelined form (kernels_reduce_pipe.hpp; cfg3
        // share: 2.72-2.85 ms against 3.03-3.07 for the workgroup-per-slice kernel); debug_flags bit 11 switches to the other of the
        // two (tests compare them)
        if constexpr (TBL * 4 > 80 * 1024) {
            if (compress && ((IPK_RPIPE_DEFAULT != 0) != ((ctx->opt_flags & 2048) != 0))) {
                constexpr int PD = IPK_RPIPE_D;
                auto kern = reduce_buckets_pipe_kernel<TBL, NT, PD>;
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(std::min<uint32_t>(n_gb, (uint32_t)ctx->num_cu)), dim3(NT), lds, ctx->stream, ctx->pool.as<uint2>(),
                                   ctx->gboff.as<uint64_t>(), ctx->clist.as<uint2>(), n_gb, NB, T, ctx->mask.as<uint32_t>(), ctx->mask_words,
                                   ctx->cvals.as<uint2>(), ctx->coff.as<uint64_t>(), ctx->rank.as<uint32_t>(), ctx->vaddr.as<uint64_t>(), ctx->ucnt.as<uint32_t>());
                HIP_TRY(ctx, hipGetLastError());
                return IPKGPU_OK;
            }
        }
        return compress ? launch(reduce_buckets_kernel<TBL, NT, true>) : launch(reduce_buckets_kernel<TBL, NT, false>);
    }
}

template <int SIGMA, int K> size_t stream_lds() {
    constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
    if constexpr (TBL == 0) return 0;
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        return TileGeo<SIGMA, K, stream_tw<SIGMA, K>()>::HEAD_BYTES + (size_t)stream_nw<SIGMA, K>() * stream_wave_scratch<SIGMA, K, CAP>() * 8 +
               (size_t)stream_nw<SIGMA, K>() * 2 * NB * SUB * 4;
    }
}
template <int SIGMA, int K> uint32_t stream_nb() {
    constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
    if constexpr (TBL == 0) return 0; else return (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
}

// ---- quad kernel: pass 1 of the stream variant for DNA k = 8..12 (kernels_quad.hpp) -----------------------
template <int SIGMA, int K> constexpr bool quad_ok() { return QuadGeo<SIGMA, K>::OK && stream_tbl<SIGMA, K>() != 0; }
// tuning knobs of the k <= 10 instantiations (build-time: -DIPK_QNW= -DIPK_QTW= -DIPK_QCAP=)
#ifndef IPK_QNW
#define IPK_QNW 4
#endif
#ifndef IPK_QTW
#define IPK_QTW 40
#endif
template <int SIGMA, int K> constexpr int quad_nw() { return K <= 10 ? IPK_QNW : IPK_QNW12; }
template <int SIGMA, int K> constexpr int quad_tw() { return K <= 10 ? IPK_QTW : IPK_QTW12; }
// the final join with one row of L per lane (kernels_quad.hpp, ROWLANE): k = 11, 12
#ifndef IPK_QROWLANE12
#define IPK_QROWLANE12 1
#endif
#ifndef IPK_QROWLANE
#define IPK_QROWLANE 0
#endif
template <int SIGMA, int K> constexpr bool quad_rowlane() { return K <= 10 ? IPK_QROWLANE != 0 : IPK_QROWLANE12 != 0; }
inline bool quad_rowlane_rt(uint32_t sigma, uint32_t k) { return sigma == 4 && (k <= 10 ? IPK_QROWLANE != 0 : IPK_QROWLANE12 != 0); }
// child nodes of one window per wavefront step (kernels_quad.hpp, ONEWIN): k = 11, 12
#ifndef IPK_QONEWIN12
#define IPK_QONEWIN12 1
#endif
template <int SIGMA, int K> constexpr bool quad_onewin() { return K >= 11 && IPK_QONEWIN12 != 0 && quad_rowlane<SIGMA, K>(); }
template <int SIGMA, int K> size_t quad_lds()
{
    if constexpr (!quad_ok<SIGMA, K>()) return 0;
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        return QuadTile<SIGMA, K, quad_tw<SIGMA, K>()>::HEAD_BYTES + (size_t)quad_nw<SIGMA, K>() * quad_wave_entries<SIGMA, K, CAP, quad_onewin<SIGMA, K>()>() * 8 +
               (size_t)quad_nw<SIGMA, K>() * NB * 8;
    }
}
template <int SIGMA, int K, bool COUNT_ONLY = false>
int launch_quad_pass1(ipkgpu_ctx* ctx, const StreamParams& sp, uint32_t n_wg)
{
    if constexpr (!quad_ok<SIGMA, K>()) { (void)sp; (void)n_wg; return fail(ctx, IPKGPU_ERR_INVALID, "quad kernel unsupported for this sigma/k"); }
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t TBL = stream_tbl<SIGMA, K>();
        constexpr int QNW = quad_nw<SIGMA, K>(), QTW = quad_tw<SIGMA, K>();
        const size_t lds = quad_lds<SIGMA, K>();
        auto kern = score_quad_kernel<SIGMA, K, CAP, QTW, QNW, TBL, COUNT_ONLY, quad_rowlane<SIGMA, K>(), quad_onewin<SIGMA, K>()>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(QNW * 64), lds, ctx->stream, sp);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}

// ---- exact-partition variant (count -> scan -> write -> reduce), kernels_score.hpp ---------------------
#ifndef IPK_XPNW
#define IPK_XPNW 11
#endif
#ifndef IPK_XPTW
#define IPK_XPTW 128
#endif
constexpr int XP_TW = IPK_XPTW;
template <int SIGMA, int K> constexpr int xp_nw() { return SIGMA == 20 ? IPK_XPNW : 11; }   // 11 waves: what fits 160 KB of LDS at AA k=6 (12 with 64-window tiles measured equal)
template <int SIGMA, int K> constexpr uint32_t xp_tbl()
{
    if (SIGMA == 20 && K == 6) return 16000u;                           // 4000 buckets per group; 64 KB reduce tables: two workgroups per CU
    if (SIGMA == 4 && K >= 13) return 32768u;                           // DNA k = 13, 14: 2048 / 8192 buckets per group (TBL a multiple of 4^7 = a row's code range)
    return stream_tbl<SIGMA, K>();
}
template <int SIGMA, int K> uint32_t xp_nb() {
    constexpr uint32_t TBL = xp_tbl<SIGMA, K>();
    if constexpr (TBL == 0) return 0; else return (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
}
template <int SIGMA, int K> size_t xp_lds() {
    constexpr uint32_t TBL = xp_tbl<SIGMA, K>();
    if constexpr (TBL == 0) return 0;
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        return TileGeo<SIGMA, K, XP_TW>::HEAD_BYTES + (size_t)xp_nw<SIGMA, K>() * stream_wave_scratch<SIGMA, K, CAP>() * 8 + (size_t)NB * 4;
    }
}
template <int SIGMA, int K, bool WRITE>
int launch_xp(ipkgpu_ctx* ctx, const XpParams& xp, uint32_t n_wg)
{
    constexpr uint32_t TBL = xp_tbl<SIGMA, K>();
    if constexpr (TBL == 0) { (void)xp; (void)n_wg; return fail(ctx, IPKGPU_ERR_INVALID, "exact-partition variant unsupported for this sigma/k"); }
    else {
        constexpr int CAP = fast_cap<SIGMA, K>();
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        constexpr size_t lds = TileGeo<SIGMA, K, XP_TW>::HEAD_BYTES + (size_t)xp_nw<SIGMA, K>() * stream_wave_scratch<SIGMA, K, CAP>() * 8 + (size_t)NB * 4;
        static_assert(lds <= 160 * 1024, "exact-partition LDS budget");
        static_assert((TileGeo<SIGMA, K, XP_TW>::HEAD_BYTES + (size_t)xp_nw<SIGMA, K>() * stream_wave_scratch<SIGMA, K, CAP>() * 8) % 8 == 0, "cursor alignment");
        auto kern = score_xp_kernel<SIGMA, K, CAP, XP_TW, xp_nw<SIGMA, K>(), TBL, WRITE>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(xp_nw<SIGMA, K>() * 64), lds, ctx->stream, xp);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}
template <int SIGMA, int K, bool WRITE>
int launch_xp_overflow(ipkgpu_ctx* ctx, const XpParams& xp)
{
    constexpr uint32_t TBL = xp_tbl<SIGMA, K>();
    if constexpr (TBL == 0 || ipow(SIGMA, K - K / 2) <= (uint32_t)fast_cap<SIGMA, K>()) {
        (void)ctx; (void)xp;
        return IPKGPU_OK;                           // lists can never overflow
    } else {
        constexpr size_t lds = TileGeo<SIGMA, K, 1>::HEAD_BYTES + (size_t)wave_scratch_entries<SIGMA, K, big_capf<SIGMA, K>()>() * 8;
        static_assert(lds + 64 <= 160 * 1024, "big-list (exact partition) LDS budget");
        auto kern = score_overflow_xp_kernel<SIGMA, K, TBL, WRITE>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(32 / OVF_NW, (160 * 1024) / (lds + 64)));
        hipLaunchKernelGGL(kern, dim3(ctx->num_cu * per_cu), dim3(OVF_NW * 64), lds, ctx->stream, xp);
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}
template <int SIGMA, int K, bool COMPRESS>
int launch_xp_reduce(ipkgpu_ctx* ctx, uint32_t n_gb, uint32_t S, uint64_t T, const uint64_t* off, uint32_t* table)
{
    constexpr uint32_t TBL = xp_tbl<SIGMA, K>();
    if constexpr (TBL == 0) { (void)n_gb; (void)S; (void)T; (void)off; (void)table; return fail(ctx, IPKGPU_ERR_INVALID, "exact-partition variant unsupported"); }
    else {
        constexpr uint32_t NB = (uint32_t)((ipow(SIGMA, K) + TBL - 1) / TBL);
        static_assert(!COMPRESS || NB == 1 || TBL % 64 == 0, "a 64-slot block must not straddle two buckets");
        constexpr int NT = TBL <= 16384 ? 512 : 1024;       // (16000-slot tables: 1024 threads measure the same 16.5 ms as 512 since the coalesced epilogue, 256 are slower: 22.3)
        constexpr size_t lds = COMPRESS ? (size_t)comp_padded_slots<TBL, NT>() * 4 + (NT / 64 + 1) * 4 : (size_t)TBL * 4;
        auto kern = reduce_ranges_kernel<TBL, NT, COMPRESS>;
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_gb), dim3(NT), lds, ctx->stream, ctx->pool.as<uint2>(), off, S, NB, T, table,
                           ctx->mask.as<uint32_t>(), ctx->mask_words, ctx->rank.as<uint32_t>(), ctx->vaddr.as<uint64_t>(), ctx->ucnt.as<uint32_t>());
        HIP_TRY(ctx, hipGetLastError());
        return IPKGPU_OK;
    }
}
template <int SIGMA, int K> uint32_t xp_tbl_value() { return xp_tbl<SIGMA, K>(); }

#define IPK_DISPATCH(SIGMA_V, K_V, EXPR_MACRO)                                         \
    do {                                                                               \
        if ((SIGMA_V) == 4) {                                                          \
            switch (K_V) {                                                             \
                case 2: EXPR_MACRO(4, 2); case 3: EXPR_MACRO(4, 3); case 4: EXPR_MACRO(4, 4); \
                case 5: EXPR_MACRO(4, 5); case 6: EXPR_MACRO(4, 6); case 7: EXPR_MACRO(4, 7); \
                case 8: EXPR_MACRO(4, 8); case 9: EXPR_MACRO(4, 9); case 10: EXPR_MACRO(4, 10); \
                case 11: EXPR_MACRO(4, 11); case 12: EXPR_MACRO(4, 12);                \
                case 13: EXPR_MACRO(4, 13); case 14: EXPR_MACRO(4, 14);                \
            }                                                                          \
        } else if ((SIGMA_V) == 20) {                                                  \
            switch (K_V) {                                                             \
                case 2: EXPR_MACRO(20, 2); case 3: EXPR_MACRO(20, 3); case 4: EXPR_MACRO(20, 4); \
                case 5: EXPR_MACRO(20, 5); case 6: EXPR_MACRO(20, 6);                  \
            }                                                                          \
        }                                                                              \
    } while (0)

uint32_t stream_buckets(uint32_t sigma, uint32_t k)
{
#define M_NB(S_, K_) return stream_nb<S_, K_>()
    IPK_DISPATCH(sigma, k, M_NB);
#undef M_NB
    return 0;
}
uint32_t stream_tbl_value(uint32_t sigma, uint32_t k)
{
#define M_TBLV(S_, K_) return stream_tbl<S_, K_>()
    IPK_DISPATCH(sigma, k, M_TBLV);
#undef M_TBLV
    return 0;
}
uint32_t stream_waves(uint32_t sigma, uint32_t k)
{
#define M_NWV(S_, K_) return (uint32_t)stream_nw<S_, K_>()
    IPK_DISPATCH(sigma, k, M_NWV);
#undef M_NWV
    return NW;
}
uint32_t stream_tile(uint32_t sigma, uint32_t k)
{
#define M_TWV(S_, K_) return (uint32_t)stream_tw<S_, K_>()
    IPK_DISPATCH(sigma, k, M_TWV);
#undef M_TWV
    return TW;
}
size_t stream_lds_bytes(uint32_t sigma, uint32_t k)
{
#define M_LDS(S_, K_) return stream_lds<S_, K_>()
    IPK_DISPATCH(sigma, k, M_LDS);
#undef M_LDS
    return 0;
}
int dispatch_stream_pass1(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const StreamParams& sp, uint32_t n_wg)
{
#define M_P1(S_, K_) return launch_stream_pass1<S_, K_>(ctx, sp, n_wg)
    IPK_DISPATCH(sigma, k, M_P1);
#undef M_P1
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
uint32_t fast_cap_value(uint32_t sigma, uint32_t k)
{
#define M_FCAP(S_, K_) return (uint32_t)fast_cap<S_, K_>()
    IPK_DISPATCH(sigma, k, M_FCAP);
#undef M_FCAP
    return 512;
}
bool quad_supported(uint32_t sigma, uint32_t k)
{
#define M_QOK(S_, K_) return quad_ok<S_, K_>()
    IPK_DISPATCH(sigma, k, M_QOK);
#undef M_QOK
    return false;
}
uint32_t quad_waves(uint32_t sigma, uint32_t k)
{
#define M_QNW(S_, K_) return (uint32_t)quad_nw<S_, K_>()
    IPK_DISPATCH(sigma, k, M_QNW);
#undef M_QNW
    return NW;
}
uint32_t quad_tile(uint32_t sigma, uint32_t k)
{
#define M_QTW(S_, K_) return (uint32_t)quad_tw<S_, K_>()
    IPK_DISPATCH(sigma, k, M_QTW);
#undef M_QTW
    return TW;
}
size_t quad_lds_bytes(uint32_t sigma, uint32_t k)
{
#define M_QLDS(S_, K_) return quad_lds<S_, K_>()
    IPK_DISPATCH(sigma, k, M_QLDS);
#undef M_QLDS
    return 0;
}
int dispatch_quad_pass1(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const StreamParams& sp, uint32_t n_wg, bool count_only = false)
{
#define M_Q1(S_, K_) return count_only ? launch_quad_pass1<S_, K_, true>(ctx, sp, n_wg) : launch_quad_pass1<S_, K_, false>(ctx, sp, n_wg)
    IPK_DISPATCH(sigma, k, M_Q1);
#undef M_Q1
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_stream_pass2(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t n_gb, uint64_t T, uint32_t* table, bool compress)
{
#define M_P2(S_, K_) return launch_stream_pass2<S_, K_>(ctx, n_gb, T, table, compress)
    IPK_DISPATCH(sigma, k, M_P2);
#undef M_P2
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_score_pos(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const ScoreParams& p)
{
#define M_SP(S_, K_) do { int rc_ = launch_score<S_, K_, true>(ctx, p); return rc_ ? rc_ : launch_overflow<S_, K_, true>(ctx, p); } while (0)
    IPK_DISPATCH(sigma, k, M_SP);
#undef M_SP
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_stream_overflow(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const StreamParams& sp)
{
#define M_SO(S_, K_) return launch_stream_overflow<S_, K_>(ctx, sp)
    IPK_DISPATCH(sigma, k, M_SO);
#undef M_SO
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_overflow(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const ScoreParams& p)
{
#define M_OV(S_, K_) return launch_overflow<S_, K_>(ctx, p)
    IPK_DISPATCH(sigma, k, M_OV);
#undef M_OV
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}

uint32_t xp_buckets(uint32_t sigma, uint32_t k)
{
#define M_XNB(S_, K_) return xp_nb<S_, K_>()
    IPK_DISPATCH(sigma, k, M_XNB);
#undef M_XNB
    return 0;
}
size_t xp_lds_bytes(uint32_t sigma, uint32_t k)
{
#define M_XLDS(S_, K_) return xp_lds<S_, K_>()
    IPK_DISPATCH(sigma, k, M_XLDS);
#undef M_XLDS
    return 0;
}
int dispatch_xp(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const XpParams& xp, uint32_t n_wg, bool write)
{
#define M_XP(S_, K_) return write ? launch_xp<S_, K_, true>(ctx, xp, n_wg) : launch_xp<S_, K_, false>(ctx, xp, n_wg)
    IPK_DISPATCH(sigma, k, M_XP);
#undef M_XP
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_xp_overflow(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, const XpParams& xp, bool write)
{
#define M_XO(S_, K_) return write ? launch_xp_overflow<S_, K_, true>(ctx, xp) : launch_xp_overflow<S_, K_, false>(ctx, xp)
    IPK_DISPATCH(sigma, k, M_XO);
#undef M_XO
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
int dispatch_xp_reduce(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t n_gb, uint32_t S, uint64_t T, const uint64_t* off, uint32_t* table,
                       bool compress)
{
#define M_XR(S_, K_) return compress ? launch_xp_reduce<S_, K_, true>(ctx, n_gb, S, T, off, table) : launch_xp_reduce<S_, K_, false>(ctx, n_gb, S, T, off, table)
    IPK_DISPATCH(sigma, k, M_XR);
#undef M_XR
    return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
}
uint32_t xp_bucket_slots(uint32_t sigma, uint32_t k)
{
#define M_XT(S_, K_) return xp_tbl_value<S_, K_>()
    IPK_DISPATCH(sigma, k, M_XT);
#undef M_XT
    return 0;
}
#ifndef IPK_KMC_RUNS_DEFAULT
#define IPK_KMC_RUNS_DEFAULT 1
#endif
#ifndef IPK_KMC_PASS
#define IPK_KMC_PASS 256          // groups per pass of the compressed key-major writer (a batch of more groups takes several passes)
#endif
#define KM_LAUNCH(KERN, CAPV, ...)                                                                                        \
    do {                                                                                                                  \
        if (P == 1) hipLaunchKernelGGL((KERN<true, CAPV>), dim3((uint32_t)(per_xcd * 8)), dim3(256), 0, ctx->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERN<false, CAPV>), dim3((uint32_t)(per_xcd * 8)), dim3(256), 0, ctx->stream, __VA_ARGS__);     \
    } while (0)

CompTable comp_table(const ipkgpu_ctx* ctx)
{
    CompTable ct;
    ct.mask = ctx->mask.as<uint32_t>(); ct.rank = ctx->rank.as<uint32_t>(); ct.vaddr = ctx->vaddr.as<uint64_t>();
    ct.pool = ctx->comp_own_vals ? ctx->cvals.as<uint2>() : ctx->pool.as<uint2>();
    ct.off = ctx->comp_own_vals ? ctx->coff.as<uint64_t>() : ctx->gboff.as<uint64_t>(); ct.mask_words = ctx->mask_words;
    ct.NB = ctx->comp_nb; ct.stride = ctx->comp_stride; ct.TBL = ctx->comp_tbl;
    return ct;
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

}  // namespace

// ---- the scoring pass shared by every output form ---------------------------------------------------
namespace {

// internal status of a scoring batch: its pair pool does not fit device memory -- the caller halves the batch and scores
// it again (the reference's answer to "does not fit" is its on-disk mode, db_builder.cpp:673-681; here: smaller batches)
constexpr int IPKGPU_RETRY_SMALLER = 100;
constexpr int IPKGPU_RETRY_AGAIN = 101;                  // internal: the same batch again with a larger pair pool (stream_batch_check)
#ifndef IPK_OVF_POOL_RATIO
#define IPK_OVF_POOL_RATIO 500
#endif

struct Plan {
    uint32_t n_mats = 0, sites = 0, sigma = 0, k = 0;
    float eps = 0;
    std::vector<uint32_t> group_ids;   // first-seen order (group_ghost_ids, db_builder.cpp:524-553)
    std::vector<uint32_t> slot_of;     // [n_mats] group index of each matrix
    uint32_t n_groups = 0, nwin = 0, tiles_per_mat = 0, chunks_per_group = 0;
    uint64_t table_size = 0, gpb = 1;  // slots per group table; groups per batch
};

int make_plan(ipkgpu_ctx* ctx, const void* logp, uint32_t n_mats, uint32_t sites, uint32_t sigma,
              const uint32_t* mat_group, uint32_t k, float log_eps, Plan& pl)
{
    if (!logp || !mat_group) return fail(ctx, IPKGPU_ERR_INVALID, "null input pointer");
    if (sigma != 4 && sigma != 20) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported alphabet size %u (4 or 20)", sigma);
    if (k < 2 || k > ipkgpu_max_k(sigma))
        return fail(ctx, IPKGPU_ERR_INVALID, "k=%u out of range [2, %u] for sigma=%u", k, ipkgpu_max_k(sigma), sigma);
    if (n_mats == 0) return fail(ctx, IPKGPU_ERR_INVALID, "no matrices");
    if (sites < k) return fail(ctx, IPKGPU_ERR_INVALID, "alignment has %u sites, fewer than k=%u", sites, k);
    pl.n_mats = n_mats; pl.sites = sites; pl.sigma = sigma; pl.k = k; pl.eps = log_eps;
    pl.slot_of.resize(n_mats);
    std::unordered_map<uint32_t, uint32_t> index;
    index.reserve(n_mats);
    for (uint32_t i = 0; i < n_mats; ++i) {
        auto it = index.find(mat_group[i]);
        if (it == index.end()) {
            it = index.emplace(mat_group[i], (uint32_t)pl.group_ids.size()).first;
            pl.group_ids.push_back(mat_group[i]);
        }
        pl.slot_of[i] = it->second;
    }
    pl.n_groups = (uint32_t)pl.group_ids.size();
    pl.table_size = ipow(sigma, (int)k);
    pl.chunks_per_group = (uint32_t)((pl.table_size + CHUNK - 1) / CHUNK);
    // groups per batch: what a group keeps resident at once -- its score table and its share of the pair pool (the scored
    // phylo-k-mers of its windows, 8 bytes each, at the rate calibrated by the context's previous call or its pre-pass; a
    // guess of 256 per window before that) -- against workspace_bytes.  An underestimate is caught later: a batch whose pool
    // does not fit is halved and scored again (IPKGPU_RETRY_SMALLER).
    pl.nwin = sites - k + 1;
    {
        const double ppw = ctx->pairs_per_window > 0 ? ctx->pairs_per_window : 256.0;
        const double pool_per_group = (double)n_mats / (double)pl.n_groups * (double)pl.nwin * ppw * 8.0 * 1.25;
        // (the exact partition ends in compressed tables, in place in the pool: occupancy bits, ranks and value addresses only -- T / 8 +
        //  T / 16 + T / 8 bytes instead of 4 T; with the dense figure 125 groups of k = 14 were scored in two batches and merged)
        const uint32_t xnb = xp_buckets(sigma, k);
        const bool no_dense = xnb != 0 && (ctx->opt_variant == 4 || (ctx->opt_variant == 0 && stream_buckets(sigma, k) == 0));
        const double per_group = (double)pl.table_size * (no_dense ? 0.3125 : 4.0) + pool_per_group;
        pl.gpb = std::max<uint64_t>(1, (uint64_t)((double)ctx->workspace_bytes / per_group));
    }
    pl.gpb = std::min<uint64_t>(pl.gpb, pl.n_groups);
    while (pl.gpb > 1 && pl.gpb * pl.chunks_per_group > 0x7fffffffull) pl.gpb /= 2;
    pl.nwin = sites - k + 1;
    pl.tiles_per_mat = (pl.nwin + TW - 1) / TW;
    return IPKGPU_OK;
}

int run_prefix(ipkgpu_ctx* ctx, const Plan& pl, const float* logp_dev)
{
    RC_TRY(ensure(ctx, ctx->best, (size_t)pl.n_mats * (pl.sites + 1) * 4));
    HIP_TRY(ctx, hipMemsetAsync(ctx->small, 0, 64, ctx->stream));
    // matrices per workgroup: as few as keep the call to one round of workgroups (the kernel's registers allow 7 per CU; 4 leaves room)
    uint32_t M = 1;
    while (M < 8 && (pl.n_mats + M - 1) / M > (uint32_t)ctx->num_cu * 4) M *= 2;
    if (ctx->opt_prefix_mats == 1 || ctx->opt_prefix_mats == 2 || ctx->opt_prefix_mats == 4 || ctx->opt_prefix_mats == 8) M = (uint32_t)ctx->opt_prefix_mats;
    const dim3 grid((pl.n_mats + M - 1) / M);
#define IPK_PREFIX(SG, MM) hipLaunchKernelGGL((prefix_max_kernel<SG, MM>), grid, dim3(256), 0, ctx->stream, logp_dev, pl.n_mats, pl.sites, ctx->best.as<float>())
    if (pl.sigma == 4) { if (M == 1) IPK_PREFIX(4, 1); else if (M == 2) IPK_PREFIX(4, 2); else if (M == 4) IPK_PREFIX(4, 4); else IPK_PREFIX(4, 8); }
    else { if (M == 1) IPK_PREFIX(20, 1); else if (M == 2) IPK_PREFIX(20, 2); else if (M == 4) IPK_PREFIX(20, 4); else IPK_PREFIX(20, 8); }
#undef IPK_PREFIX
    HIP_TRY(ctx, hipGetLastError());
    return IPKGPU_OK;
}

int scan_u32(ipkgpu_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out);

// Exact-partition variant of one batch (kernels_score.hpp): count -> scan -> write -> reduce -> big-list windows.
// ctx->gm holds the batch's group -> matrices CSR; p carries the shared counters and the big-list queue.
int score_batch_xp(ipkgpu_ctx* ctx, const Plan& pl, const float* logp_dev, uint32_t gb, uint32_t nb, ScoreParams& p, uint32_t XNB, bool compress)
{
    (void)nb;
    const uint32_t tiles_per_mat = (pl.nwin + XP_TW - 1) / XP_TW;
    const size_t lds_bytes = xp_lds_bytes(pl.sigma, pl.k);
    const uint64_t wg_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(32 / (pl.sigma == 20 ? IPK_XPNW : 11), (160 * 1024) / std::max<size_t>(lds_bytes, 1)));
    const uint64_t slots = (uint64_t)ctx->num_cu * wg_per_cu;
    // four rounds of resident workgroups balance the tail; a unit costs only its NB counters
    const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((slots * 4) / gb, tiles_per_mat));
    const uint64_t n_gb = (uint64_t)gb * XNB;
    const uint32_t stride = S + 1;                           // slot S of every (group, bucket): the big-list windows
    const uint64_t n_units = n_gb * stride;
    if (n_gb > 0x7fffffffull || (uint64_t)gb * S > 0x7fffffffull) return fail(ctx, IPKGPU_ERR_INVALID, "batch too large for one launch");
    RC_TRY(ensure(ctx, ctx->gbcnt, n_units * 4));
    RC_TRY(ensure(ctx, ctx->gboff, (n_units + 1) * 8));
    RC_TRY(ensure(ctx, ctx->gbcur, n_gb * 4));
    HIP_TRY(ctx, hipMemsetAsync(ctx->gbcnt.p, 0, n_units * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->gbcur.p, 0, n_gb * 4, ctx->stream));

    XpParams xp;
    StreamParams& sp = xp.s;
    sp.logp = logp_dev; sp.best = ctx->best.as<float>();
    sp.gm_off = ctx->gm.as<uint32_t>(); sp.gm_list = ctx->gm.as<uint32_t>() + gb + 1;
    sp.sites = pl.sites; sp.nwin = pl.nwin; sp.tiles_per_mat = tiles_per_mat; sp.S = S;
    sp.eps = pl.eps;
    sp.pool = nullptr; sp.pool_cap = 0; sp.pool_next = nullptr; sp.desc = nullptr; sp.pool_ovf = nullptr; sp.pre_chunks = 0;
    sp.emitted = p.emitted; sp.ovf_queue = p.ovf_queue; sp.ovf_count = p.ovf_count; sp.mat_slot = p.mat_slot; sp.big_ovf = p.big_ovf;
    sp.flags = (uint32_t)(ctx->opt_flags) & 4u;      // (bit 2: list building only -- timing experiments)
    xp.cnt = ctx->gbcnt.as<uint32_t>();
    xp.off = ctx->gboff.as<uint64_t>();
    xp.stride = stride;
    xp.ovcur = ctx->gbcur.as<uint32_t>();
    xp.start = nullptr;

    Stopwatch sw(ctx->stream, &ctx->events);
    const int ev_a = sw.mark();
    RC_TRY(dispatch_xp(ctx, pl.sigma, pl.k, xp, gb * S, false));
    const int ev_a2 = sw.mark();
    RC_TRY(dispatch_xp_overflow(ctx, pl.sigma, pl.k, xp, false));          // reads the queue length on the device
    RC_TRY(scan_u32(ctx, xp.cnt, n_units, ctx->gboff.as<uint64_t>()));
    uint64_t total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->gboff.as<uint64_t>() + n_units, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const bool over_limit = ctx->opt_pool_limit > 0 && total * 8 > (uint64_t)ctx->opt_pool_limit;
        if (over_limit || (total * 8 > ctx->pool.cap && total * 8 > free_b + ctx->pool.cap)) {
            if (gb > 1) return IPKGPU_RETRY_SMALLER;
            return fail(ctx, IPKGPU_ERR_NOMEM, "the pair pool of ONE branch group (%llu pairs) does not fit device memory", (unsigned long long)total);
        }
    }
    RC_TRY(ensure(ctx, ctx->pool, std::max<uint64_t>(total, 1) * 8 + 256));     // (+256: km_write_c_kernel reads up to 32 values from a row's start)
    {
        // the write pass' cursors: every workgroup's NB starting offsets side by side, relative to its group's first pair
        const uint64_t n_start = (uint64_t)gb * S * XNB;
        RC_TRY(ensure(ctx, ctx->xstart, n_start * 4));
        hipLaunchKernelGGL(xp_unit_starts_kernel, dim3((uint32_t)((n_start + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->gboff.as<uint64_t>(), XNB, S, stride, n_start, ctx->xstart.as<uint32_t>(),
                           reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 56));
        HIP_TRY(ctx, hipGetLastError());
        xp.start = ctx->xstart.as<uint32_t>();
    }
    sp.pool = ctx->pool.as<uint2>();
    const int ev_c = sw.mark();
    RC_TRY(dispatch_xp(ctx, pl.sigma, pl.k, xp, gb * S, true));
    const int ev_d0 = sw.mark();
    RC_TRY(dispatch_xp_overflow(ctx, pl.sigma, pl.k, xp, true));
    const int ev_d = sw.mark();
    if (compress) {
        RC_TRY(ensure(ctx, ctx->rank, (size_t)gb * (ctx->mask_words / 2) * 4));
        RC_TRY(ensure(ctx, ctx->vaddr, (size_t)gb * (ctx->mask_words / 2) * 8));
        RC_TRY(ensure(ctx, ctx->ucnt, n_gb * 4));
    } else {
        RC_TRY(ensure(ctx, ctx->table, (size_t)gb * pl.table_size * 4));
    }
    RC_TRY(dispatch_xp_reduce(ctx, pl.sigma, pl.k, (uint32_t)n_gb, stride, pl.table_size, ctx->gboff.as<uint64_t>(),
                              compress ? nullptr : ctx->table.as<uint32_t>(), compress));
    const int ev_e = sw.mark();
    uint32_t too_big = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&too_big, reinterpret_cast<char*>(ctx->small) + 56, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (too_big) return fail(ctx, IPKGPU_ERR_INVALID, "a branch group scores 2^32 phylo-k-mers or more in one batch (exact-partition variant: 32-bit offsets inside a group)");
    ctx->mask_valid = true;
    ctx->table_compressed = compress;
    ctx->comp_nb = XNB; ctx->comp_stride = stride; ctx->comp_tbl = xp_bucket_slots(pl.sigma, pl.k);
    ctx->acc_main_ms += sw.ms(ev_a, ev_a2) + sw.ms(ev_c, ev_d0);
    ctx->acc_count_ms += sw.ms(ev_a, ev_a2); ctx->acc_write_ms += sw.ms(ev_c, ev_d0);
    ctx->acc_reduce_ms += sw.ms(ev_d, ev_e);
    ctx->main_kernel = "score_xp_kernel";
    return IPKGPU_OK;
}

// What a stream-variant batch owes after the wait that made ctx->h_rb valid: the pool's state, the statistics, the calibration of
// the next call.  IPKGPU_RETRY_AGAIN: the pool ran out -- the batch is to be scored again (ctx->pool_want_min says how large).
int stream_batch_check(ipkgpu_ctx* ctx)
{
    ipkgpu_ctx::Pending& pd = ctx->pend;
    if (!pd.active) return IPKGPU_OK;
    pd.active = false;
    std::unique_ptr<Stopwatch> sw(pd.sw.release());
    const uint32_t n_ovf = ctx->h_rb[4], pool_exhausted = ctx->h_rb[9];
    if (pool_exhausted) {
        if (pd.cap >= pd.max_chunks) {
            ctx->pool_want_min = 0;
            if (pd.gb > 1) return IPKGPU_RETRY_SMALLER;
            return fail(ctx, IPKGPU_ERR_NOMEM, "the pair pool of ONE branch group is exhausted at the device-memory limit");
        }
        ctx->pool_want_min = pd.cap * 2;
        return IPKGPU_RETRY_AGAIN;
    }
    ctx->pool_want_min = 0;
    memcpy(&ctx->emitted_host, ctx->h_rb, 8);
    ctx->emitted_fetched = true;
    // the next call skips the wait after pass 1 if its big-list windows would take the atomic kernel anyway
    ctx->spec_skip_wait = !(n_ovf > 0 && (uint64_t)n_ovf * IPK_OVF_POOL_RATIO >= pd.windows && pd.pool_ovf_ok);
    if (sw) { ctx->acc_main_ms += sw->ms(pd.ev_a, pd.ev_b); ctx->acc_reduce_ms += sw->ms(pd.ev_c, pd.ev_d); }
    return IPKGPU_OK;
}

// Scores groups [g0, g0 + gb) into ctx->table ([gb][table_size]); *emitted_out = scored phylo-k-mers of the batch.
// defer: a stream-variant batch may return with its last wait still owed (ctx->pend.active) -- the caller waits on the stream
// later anyway and then calls score_batch_finish.
int score_batch_impl(ipkgpu_ctx* ctx, const Plan& pl, const float* logp_dev, uint32_t g0, uint32_t gb,
                     std::vector<uint32_t>& idx_host, bool defer)
{
    const uint32_t n_mats = pl.n_mats;
    (void)idx_host;
    // host side of the batch's index arrays, in pinned staging: [mat_list | mat_slot] (2 x n_mats) and the group -> matrices
    // CSR [gb + 1 offsets | matrices]; both uploads leave without a wait
    uint32_t nb = 0;
    for (uint32_t i = 0; i < n_mats; ++i) nb += (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb);
    const size_t gm_words = (size_t)gb + 1 + nb;
    void* stage = nullptr;
    RC_TRY(upload_stage(ctx, ((size_t)n_mats * 2 + gm_words) * 4, &stage));
    uint32_t* mat_list_h = reinterpret_cast<uint32_t*>(stage);
    uint32_t* mat_slot_h = mat_list_h + n_mats;
    uint32_t* gm = mat_slot_h + n_mats;
    nb = 0;
    for (uint32_t i = 0; i < n_mats; ++i) {
        mat_slot_h[i] = 0;
        if (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb) { mat_list_h[nb++] = i; mat_slot_h[i] = pl.slot_of[i] - g0; }
    }
    for (uint32_t i = nb; i < n_mats; ++i) mat_list_h[i] = 0;
    // matrices of each group of the batch (CSR)
    for (size_t i = 0; i < gm_words; ++i) gm[i] = 0;
    for (uint32_t i = 0; i < n_mats; ++i)
        if (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb) gm[pl.slot_of[i] - g0 + 1]++;
    for (uint32_t g = 0; g < gb; ++g) gm[g + 1] += gm[g];
    {
        std::vector<uint32_t> cur(gm, gm + gb);
        for (uint32_t i = 0; i < n_mats; ++i)
            if (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb) gm[(size_t)gb + 1 + cur[pl.slot_of[i] - g0]++] = i;
    }
    RC_TRY(ensure(ctx, ctx->idx, (size_t)n_mats * 8));
    RC_TRY(ensure(ctx, ctx->ovfq, (size_t)nb * pl.nwin * 8));
    RC_TRY(ensure(ctx, ctx->gm, gm_words * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->idx.p, mat_list_h, (size_t)n_mats * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->gm.p, gm, gm_words * 4, hipMemcpyHostToDevice, ctx->stream));
    RC_TRY(upload_staged(ctx));

    ScoreParams p;
    p.logp = logp_dev;
    p.best = ctx->best.as<float>();
    p.mat_list = ctx->idx.as<uint32_t>();
    p.mat_slot = ctx->idx.as<uint32_t>() + n_mats;
    p.n_batch_mats = nb; p.sites = pl.sites; p.nwin = pl.nwin; p.tiles_per_mat = pl.tiles_per_mat;
    p.eps = pl.eps;
    p.table = ctx->table.p;
    p.mat_rank = nullptr;
    p.table_size = pl.table_size;
    p.emitted = reinterpret_cast<unsigned long long*>(ctx->small);
    p.ovf_queue = ctx->ovfq.as<unsigned long long>();
    p.ovf_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 16);
    p.flags = (uint32_t)(ctx->opt_variant == 99 ? 1 : 0);
    p.mask = nullptr; p.mask_words = 0;
    // capped big-list capacity (DNA k >= 13, kernels_score.hpp big_capf): the flag word, checked at the end of the batch
    const bool capped_lists = pl.sigma == 4 && pl.k >= 13;
    p.big_ovf = capped_lists ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 44) : nullptr;
    if (capped_lists) HIP_TRY(ctx, hipMemsetAsync(p.big_ovf, 0, 4, ctx->stream));
    auto check_capped = [&]() -> int {
        if (!capped_lists) return IPKGPU_OK;
        uint32_t hit = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&hit, p.big_ovf, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (hit) return fail(ctx, IPKGPU_ERR_INVALID, "a window's half list exceeds %d entries: beyond the capacity of this engine at k = %u "
                                                      "(the lists of k >= 13 are capped; lower omega's reach or k)", BIG_CAP_ENTRIES, pl.k);
        return IPKGPU_OK;
    };
    ctx->mask_valid = false;
    ctx->table_compressed = false;
    const uint32_t NBK = stream_buckets(pl.sigma, pl.k);
    const uint32_t XNB = xp_buckets(pl.sigma, pl.k);
    // variant 0 = default: stream where its per-wave chunk state fits (all DNA k, AA k <= 5), exact partition for
    // AA k=6; 1 = global atomics, 2 = stream (diagnostic flags honoured), 3 = exact partition wherever it exists
    //                 4 = exact partition ending in the compressed table form (the default for AA k=6: no dense tables --
    //                     64 GB less at cfg4, 40 % fewer bytes moved; 90.5 vs 92.0 ms through dense tables)
    const bool use_xp = XNB != 0 && (ctx->opt_variant == 3 || ctx->opt_variant == 4 || (ctx->opt_variant == 0 && NBK == 0));
    const bool xp_compress = use_xp && ctx->opt_variant != 3;
    const bool use_stream = !use_xp && NBK != 0 && NBK <= 2048 && (ctx->opt_variant == 0 || ctx->opt_variant == 2 || ctx->opt_variant == 5 ||
                                                                   ctx->opt_variant == 6 || ctx->opt_variant == 7);
    // pass 1 of the stream variant: the quad kernel (kernels_quad.hpp) where it exists (DNA k = 8..12), variant 2 forces the
    // first-generation score_stream_kernel, variant 5 asks for the quad kernel explicitly
    const bool use_quad = use_stream && ctx->opt_variant != 2 && quad_supported(pl.sigma, pl.k);
    if (!use_stream) {                                     // (the stream variant resets its counters in one launch: small_reset_kernel)
        HIP_TRY(ctx, hipMemsetAsync(ctx->small, 0, 8, ctx->stream));            // per-batch scored-k-mer counter
        HIP_TRY(ctx, hipMemsetAsync(p.ovf_count, 0, 4, ctx->stream));
    }
    // The chunk-fed reduce ends in the compressed table form (occupancy bits + rank + the non-empty slots' score codes, comp_table.hpp)
    // where the scored k-mers fill the key space sparsely: DNA k = 11, 12 (cfg3: 38 % of 4^12 slots per group -- no 64 MB dense table
    // per group to write and read back); variant 6 forces it for any stream (sigma, k), variant 7 forces dense tables.  The big-list
    // windows then always go through the pool (the atomic kernel needs dense tables), which bounds the field widths of their queue.
    const bool pool_ovf_ok = gb < (1u << 22) && n_mats < (1u << 21) && pl.nwin < (1u << 21);
    bool s_compress = use_stream && pool_ovf_ok && ctx->opt_variant != 7 && ctx->opt_variant != 2 &&
                      (ctx->opt_variant == 6 || (ctx->opt_variant == 0 && stream_tbl_value(pl.sigma, pl.k) == 32768u && pl.sigma == 4));
    ctx->comp_own_vals = false;
    const uint32_t SNW = use_quad ? quad_waves(pl.sigma, pl.k) : stream_waves(pl.sigma, pl.k);
    const uint32_t STW = use_quad ? quad_tile(pl.sigma, pl.k) : stream_tile(pl.sigma, pl.k);
    const uint32_t s_tiles_per_mat = (pl.nwin + STW - 1) / STW;
    if (!use_stream && !xp_compress) {                     // (the stream variant decides its table form below, once the pair rate is known)
        RC_TRY(ensure(ctx, ctx->table, (size_t)gb * pl.table_size * 4));
        p.table = ctx->table.p;
    }
    if (!use_stream && !use_xp) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->table.p, 0, (size_t)gb * pl.table_size * 4, ctx->stream));
        Stopwatch sw(ctx->stream, &ctx->events);
        const int a = sw.mark();
        RC_TRY(dispatch_score(ctx, pl.sigma, pl.k, p));
        const int b = sw.mark();
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->acc_main_ms += sw.ms(a, b);
        ctx->main_kernel = "score_tiles_kernel";
        return check_capped();
    }

    // ---- stream variant: pass 1 (append pairs) -> chunk index -> pass 2 (LDS reduce) -> big-list windows
    // occupancy bits of the tables: written by the LDS reduce pass, kept current by the big-list kernel, read by km_count
    ctx->mask_words = 2 * ((pl.table_size + 63) / 64);
    RC_TRY(ensure(ctx, ctx->mask, (size_t)gb * ctx->mask_words * 4));
    p.mask = ctx->mask.as<uint32_t>(); p.mask_words = ctx->mask_words;
    if (use_xp) {
        const int rc = score_batch_xp(ctx, pl, logp_dev, gb, nb, p, XNB, xp_compress);
        return rc ? rc : check_capped();
    }

    // Segments (workgroups) per group.  More workgroups balance the tail of the persistent kernel, but every
    // wavefront keeps one open chunk per key bucket, so workgroups x waves x buckets must stay well below
    // the number of chunks the data itself fills (k = 12 has 512 buckets).
    const uint64_t windows = (uint64_t)nb * pl.nwin;
    // First call of a context: the pool is sized from a count-only pre-pass over a sample of the batch's groups (every
    // stride-th one) instead of a worst-case guess -- cfg2: 24 GB instead of 51 GB to allocate, and no redo loop.
    if (ctx->pairs_per_window <= 0) {
        const uint32_t n_s = std::min<uint32_t>(gb, std::max<uint32_t>(2, gb / 64));
        const uint32_t stride_g = gb / n_s;
        std::vector<uint32_t> sgm((size_t)n_s + 1, 0), slist;
        for (uint32_t j = 0; j < n_s; ++j) {
            const uint32_t g = j * stride_g;
            for (uint32_t m = gm[g]; m < gm[g + 1]; ++m) slist.push_back(gm[(size_t)gb + 1 + m]);
            sgm[j + 1] = (uint32_t)slist.size();
        }
        std::vector<uint32_t> both(sgm);
        both.insert(both.end(), slist.begin(), slist.end());
        RC_TRY(ensure(ctx, ctx->tmp_a, both.size() * 4));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->tmp_a.p, both.data(), both.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(p.emitted, 0, 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(p.ovf_count, 0, 4, ctx->stream));
        StreamParams ss;
        ss.logp = logp_dev; ss.best = ctx->best.as<float>();
        ss.gm_off = ctx->tmp_a.as<uint32_t>(); ss.gm_list = ctx->tmp_a.as<uint32_t>() + n_s + 1;
        ss.sites = pl.sites; ss.nwin = pl.nwin; ss.tiles_per_mat = s_tiles_per_mat;
        ss.S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(s_tiles_per_mat, (uint64_t)ctx->num_cu * 4 / n_s));
        ss.eps = pl.eps;
        ss.pool = nullptr; ss.pool_cap = 0; ss.pool_next = nullptr; ss.desc = nullptr; ss.pool_ovf = nullptr; ss.pre_chunks = 0;
        ss.emitted = p.emitted; ss.ovf_queue = p.ovf_queue; ss.ovf_count = p.ovf_count; ss.mat_slot = p.mat_slot;
        ss.flags = 2u;
        if (use_quad) RC_TRY(dispatch_quad_pass1(ctx, pl.sigma, pl.k, ss, n_s * ss.S, true));
        else RC_TRY(dispatch_stream_pass1(ctx, pl.sigma, pl.k, ss, n_s * ss.S));
        unsigned long long se = 0; uint32_t so = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&se, p.emitted, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&so, p.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const uint64_t s_windows = (uint64_t)slist.size() * pl.nwin;
        // windows that went to the big-list queue were not counted: assume they carry CAP^2 / 4 pairs each (generous)
        const double cap_pairs = 0.25 * (double)fast_cap_value(pl.sigma, pl.k) * (double)fast_cap_value(pl.sigma, pl.k);
        if (s_windows) ctx->pairs_per_window = std::max(1.0, ((double)se + (double)so * cap_pairs) / (double)s_windows * 1.15);
    }
    const double ppw_est = ctx->pairs_per_window > 0 ? ctx->pairs_per_window : 256.0;
    // ... and wherever the groups are small next to the key space: short alignments (cfg5's D652-like shape: 2 x 1391 windows x 133 pairs
    // per group against 4^10 slots, a third of them ever touched) leave dense tables mostly empty, and the reduce would write and the
    // key-major writer read 4 MB per group for 1.4 MB of scores
    if (!s_compress && use_stream && pool_ovf_ok && ctx->opt_variant == 0 && pl.table_size >= (1u << 16) &&
        (double)windows * ppw_est < 0.5 * (double)gb * (double)pl.table_size)
        s_compress = true;
    if (!s_compress) {
        RC_TRY(ensure(ctx, ctx->table, (size_t)gb * pl.table_size * 4));
        p.table = ctx->table.p;
    }
    const size_t lds_bytes = use_quad ? quad_lds_bytes(pl.sigma, pl.k) : stream_lds_bytes(pl.sigma, pl.k);
    const uint64_t wg_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(32 / SNW, (160 * 1024) / std::max<size_t>(lds_bytes, 1)));
    const uint64_t slots = (uint64_t)ctx->num_cu * wg_per_cu;
    const uint32_t CH = chunk_pairs_rt(stream_tbl_value(pl.sigma, pl.k));      // pairs per chunk of this (sigma, k)
    const uint64_t expected_chunks = (uint64_t)((double)windows * ppw_est / CH);
#ifndef IPK_OVF_POOL_RATIO
#define IPK_OVF_POOL_RATIO 500
#endif
#ifndef IPK_WG_CHUNKS2
#define IPK_WG_CHUNKS2 2      // a wavefront should fill at least IPK_WG_CHUNKS2 / 2 chunks per bucket (4: +3.5 % at a 125-group share of cfg2)
#endif
    const uint64_t wg_chunks2 = ctx->opt_wg_chunks2 > 0 ? (uint64_t)ctx->opt_wg_chunks2 : (uint64_t)IPK_WG_CHUNKS2;
    const uint64_t max_wg = std::max<uint64_t>(slots, expected_chunks * 2 / wg_chunks2 / ((uint64_t)SNW * NBK * SUB));
    // whole rounds of resident workgroups: a partial last round leaves CUs idle for a full workgroup's run time
#ifndef IPK_ROUNDS
#define IPK_ROUNDS 8
#endif
    const uint64_t rounds = std::max<uint64_t>(1, std::min<uint64_t>(ctx->opt_rounds > 0 ? (uint64_t)ctx->opt_rounds : (uint64_t)IPK_ROUNDS, max_wg / slots));
    uint64_t S64 = std::max<uint64_t>(1, (slots * rounds) / gb);
    uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(S64, s_tiles_per_mat));
    const uint32_t n_wg = gb * S;
    const uint64_t n_waves = (uint64_t)n_wg * SNW;
    const uint64_t n_gb = (uint64_t)gb * NBK;

    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    uint64_t max_chunks = std::min<uint64_t>(0xFFFFFFF0ull, (uint64_t)(free_b + ctx->pool.cap + ctx->desc.cap) * 9 / 10 / (CH * 8 + 8));
    if (ctx->opt_pool_limit > 0) max_chunks = std::min<uint64_t>(max_chunks, (uint64_t)ctx->opt_pool_limit / (CH * 8 + 8));
    // pool size: pairs expected (calibrated by the previous call, +25 %) plus every wave's open chunks and id batches.
    // An existing pool is kept as long as it covers the expectation without the margin -- regrowing a multi-GB
    // buffer costs hundreds of ms, and an underestimate is caught by the redo path anyway.
    const double ppw = ctx->pairs_per_window > 0 ? ctx->pairs_per_window : 256.0;
    const uint64_t ovf_waves = (uint64_t)ctx->num_cu * 4 * OVF_NW;          // big-list kernel: its waves hold open chunks too
    const uint64_t slack = 2 * (n_waves + ovf_waves) * NBK * SUB + (n_waves + ovf_waves) * ALLOC_BATCH + 1024;
    uint64_t want = (uint64_t)((double)windows * ppw * 1.25 / CH) + slack;
    {
        // (the pool holds one spare chunk past its last id: kernels_quad.hpp, RowAppender)
        const uint64_t have = std::min<uint64_t>(std::max<uint64_t>(ctx->pool.cap / (CH * 8), 1) - 1, ctx->desc.cap / 8);
        // expectation without the safety margins: pairs +10 %, ONE open chunk per (wave, bucket), the id batches
        const uint64_t need = (uint64_t)((double)windows * ppw * 1.1 / CH) + (n_waves + ovf_waves) * (NBK * SUB + ALLOC_BATCH) + 1024;
        if (have >= need) want = have;
    }
    want = std::max<uint64_t>(want, ctx->pool_want_min);           // (the redo of a batch whose pool ran out, found at a deferred check)
    for (int attempt = 0; attempt < 6; ++attempt) {
        uint64_t cap = std::min<uint64_t>(want, max_chunks);
        const bool forced = attempt == 0 && ctx->opt_pool_chunks > 0 && ctx->pool_want_min == 0;
        if (forced) cap = (uint64_t)ctx->opt_pool_chunks;
        else if (cap < n_waves * NBK * SUB) {
            if (gb > 1) return IPKGPU_RETRY_SMALLER;
            return fail(ctx, IPKGPU_ERR_NOMEM, "the pair pool of ONE branch group does not fit device memory");
        }
        RC_TRY(ensure(ctx, ctx->pool, (cap + 1) * CH * 8));
        RC_TRY(ensure(ctx, ctx->desc, cap * 8));
        if (!forced) cap = std::min<uint64_t>(ctx->pool.cap / (CH * 8) - 1, ctx->desc.cap / 8);
        RC_TRY(ensure(ctx, ctx->gbcnt, 2 * n_gb * 4));             // [chunks per (group, bucket) | scatter cursors]: one fill for both
        uint32_t* const d_gbcur = ctx->gbcnt.as<uint32_t>() + n_gb;
        RC_TRY(ensure(ctx, ctx->gboff, (n_gb + 1) * 8));
        uint32_t* d_pool_next = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 32);
        uint32_t* d_pool_ovf = d_pool_next + 1;
        // (row-per-lane quad kernel: the wavefronts' first chunks are handed out by position, the counter starts behind them)
        // (the candidate-per-lane kernel stores through 32-bit offsets from a base chunk, at first the wavefront's own pre-assigned
        //  chunks; in a pool beyond 4 GiB the first chunk it DRAWS lies outside that window and the wavefront rebases once -- by then
        //  its buckets have filled most of their first chunks without a single roll; debug_flags bit 5 switches the hand-out off)
        const bool pre_ok = use_quad && n_waves * NBK + 1024 <= cap && !(ctx->opt_flags & 32);
        const uint32_t pre_chunks = pre_ok ? (uint32_t)(n_waves * NBK) : 0u;
        // (one launch for the batch's counters -- a retry must not double count -- and the chunk counter's start)
        // (and the groups' tile counters of the quad kernel: its workgroups draw their tiles; debug_flags bit 7: fixed ranges)
        uint32_t* d_tile_next = nullptr;
        // (only where a workgroup has at least four tiles to its name: with fewer there is nothing to balance and the draws cost)
        if (use_quad && !(ctx->opt_flags & 128) && ((uint64_t)nb * s_tiles_per_mat >= 4ull * gb * S || (ctx->opt_flags & 256))) {
            RC_TRY(ensure(ctx, ctx->tile_next, (size_t)gb * 4));
            d_tile_next = ctx->tile_next.as<uint32_t>();
        }
        hipLaunchKernelGGL(small_reset_kernel, dim3((std::max<uint32_t>(gb, 14) + 255) / 256), dim3(256), 0, ctx->stream,
                           reinterpret_cast<uint32_t*>(ctx->small), pre_chunks, d_tile_next, gb, S);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemsetAsync(ctx->desc.p, 0, cap * 8, ctx->stream));   // ids drawn but never opened stay empty

        StreamParams sp;
        sp.logp = logp_dev; sp.best = ctx->best.as<float>();
        sp.gm_off = ctx->gm.as<uint32_t>(); sp.gm_list = ctx->gm.as<uint32_t>() + gb + 1;
        sp.sites = pl.sites; sp.nwin = pl.nwin; sp.tiles_per_mat = s_tiles_per_mat; sp.S = S;
        sp.eps = pl.eps;
        sp.pool = ctx->pool.as<uint2>(); sp.pool_cap = (uint32_t)cap; sp.pool_next = d_pool_next;
        sp.desc = ctx->desc.as<unsigned long long>(); sp.pool_ovf = d_pool_ovf;
        sp.emitted = p.emitted; sp.ovf_queue = p.ovf_queue; sp.ovf_count = p.ovf_count; sp.mat_slot = p.mat_slot;
        sp.flags = (uint32_t)(ctx->opt_flags);
        sp.pre_chunks = pre_chunks;
        sp.tile_next = d_tile_next;
        std::unique_ptr<Stopwatch> sw_own(new Stopwatch(ctx->stream, &ctx->events));
        Stopwatch& sw = *sw_own;
        const int ev_a = sw.mark();
        if (use_quad) RC_TRY(dispatch_quad_pass1(ctx, pl.sigma, pl.k, sp, n_wg));
        else RC_TRY(dispatch_stream_pass1(ctx, pl.sigma, pl.k, sp, n_wg));
        const int ev_b = sw.mark();
        // windows whose half lists overflowed the fast path: big-list kernel.  With the queue sorted by group it
        // appends to the same pool (LDS max-reduce in pass 2); otherwise (field widths exceeded) it falls back to
        // global atomics on the finished tables after pass 2.
        // (one host round trip for the queue length and the pool's state: they are read again below only if the big-list windows
        //  go through the pool, which draws chunks too)
        // A call that follows one with few big-list windows does not wait here at all (spec): chunk index and reduce take the
        // number of chunks from the device, the big-list windows take the atomic kernel (always correct on dense tables; it reads
        // the queue length itself), and the pool's state is checked with the call's last wait -- an exhausted pool then costs the
        // redo of the reduce as well, which the calibrated pool size makes rare.
        const bool spec = ctx->spec_skip_wait && !s_compress && !(ctx->opt_flags & 64);
        uint32_t n_ovf = 0;
        uint32_t h[2] = {0, 0};
        if (!spec) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rb, ctx->small, 64, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            n_ovf = ctx->h_rb[4]; h[0] = ctx->h_rb[8]; h[1] = ctx->h_rb[9];
        }
        // a handful of big-list windows (< 0.2 % of the batch) are cheaper through the atomic kernel after pass 2 than
        // through sort + pool kernel; flat posteriors put a large share of the pairs there and need the pool
        const bool ovf_in_pool = n_ovf > 0 && ((uint64_t)n_ovf * IPK_OVF_POOL_RATIO >= windows || s_compress) && pool_ovf_ok;
        if (ovf_in_pool) {
            RC_TRY(ensure(ctx, ctx->tmp_a, (size_t)n_ovf * 8));
            RC_TRY(ensure(ctx, ctx->tmp_b, (size_t)n_ovf * 8));
            hipLaunchKernelGGL(ovf_group_keys_kernel, dim3((n_ovf + 255) / 256), dim3(256), 0, ctx->stream,
                               p.ovf_queue, n_ovf, p.mat_slot, ctx->tmp_a.as<unsigned long long>());
            HIP_TRY(ctx, hipGetLastError());
            size_t tmp_bytes = 0;
            HIP_TRY(ctx, rocprim::radix_sort_keys(nullptr, tmp_bytes, ctx->tmp_a.as<unsigned long long>(), ctx->tmp_b.as<unsigned long long>(),
                                                  (size_t)n_ovf, 0, 64, ctx->stream));
            RC_TRY(ensure(ctx, ctx->tmp_c, tmp_bytes));
            HIP_TRY(ctx, rocprim::radix_sort_keys(ctx->tmp_c.p, tmp_bytes, ctx->tmp_a.as<unsigned long long>(), ctx->tmp_b.as<unsigned long long>(),
                                                  (size_t)n_ovf, 0, 64, ctx->stream));
            StreamParams so = sp;
            so.ovf_queue = ctx->tmp_b.as<unsigned long long>();
            // with the quad kernel the pool's pairs are counted from the chunk descriptors: the big-list kernel's own count goes nowhere
            if (use_quad) so.emitted = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ctx->small) + 48);
            RC_TRY(dispatch_stream_overflow(ctx, pl.sigma, pl.k, so));
        }
        if (ovf_in_pool) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rb, ctx->small, 64, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            h[0] = ctx->h_rb[8]; h[1] = ctx->h_rb[9];
        }
        if (h[1] != 0) {                    // pool exhausted: h[0] chunks were asked for
            if (cap >= max_chunks) {
                if (gb > 1) return IPKGPU_RETRY_SMALLER;
                return fail(ctx, IPKGPU_ERR_NOMEM, "the pair pool of ONE branch group is exhausted at the device-memory limit");
            }
            // (h[0] = ids drawn is no measure of the need: an exhausted pool is asked again at every append)
            want = std::max<uint64_t>(cap * 2, want);
            continue;
        }
        const uint32_t n_used = spec ? (uint32_t)cap : (uint32_t)std::min<uint64_t>(h[0], cap);   // (spec: the kernels stop at the device's count)
        const uint32_t* n_used_dev = spec ? d_pool_next : nullptr;
        // chunk index: chunk ids grouped by (group, bucket).  Unused/never-closed chunks have count 0 and are skipped;
        // descriptors of this batch's chunks were all written by pass 1 (closed or flushed), stale ones are overwritten
        // or lie beyond n_used.
        HIP_TRY(ctx, hipMemsetAsync(ctx->gbcnt.p, 0, 2 * n_gb * 4, ctx->stream));
        RC_TRY(ensure(ctx, ctx->clist, ((uint64_t)n_used + 16) * 8));   // (+16: the persistent reduce fetches a trip's descriptors with one scalar load)
        if (s_compress) {
            RC_TRY(ensure(ctx, ctx->croom, 2 * n_gb * 4));                  // [pairs per (group, bucket) | room of its values]
            HIP_TRY(ctx, hipMemsetAsync(ctx->croom.p, 0, n_gb * 4, ctx->stream));
        }
        if (n_used) {
            hipLaunchKernelGGL(chunk_hist_kernel, dim3(std::min<uint32_t>((n_used + 255) / 256, 2048u)), dim3(256), 0, ctx->stream,
                               ctx->desc.as<unsigned long long>(), n_used, ctx->gbcnt.as<uint32_t>(),
                               use_quad ? p.emitted : (unsigned long long*)nullptr, s_compress ? ctx->croom.as<uint32_t>() : (uint32_t*)nullptr,
                               n_used_dev);
            HIP_TRY(ctx, hipGetLastError());
        }
        RC_TRY(scan_u32(ctx, ctx->gbcnt.as<uint32_t>(), n_gb, ctx->gboff.as<uint64_t>()));
        if (s_compress) {
            // where the slices' values go: room for min(pairs, slots) values per (group, bucket), scanned; the buffer itself is sized
            // from what the host knows without another wait -- no more values than pairs, no more pairs than the used chunks hold
            const uint32_t TBLv = stream_tbl_value(pl.sigma, pl.k);
            hipLaunchKernelGGL(comp_slice_room_kernel, dim3((uint32_t)((n_gb + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->croom.as<uint32_t>(), (uint32_t)n_gb, NBK, pl.table_size, TBLv, ctx->croom.as<uint32_t>() + n_gb);
            HIP_TRY(ctx, hipGetLastError());
            RC_TRY(ensure(ctx, ctx->coff, (n_gb + 1) * 8));
            RC_TRY(scan_u32(ctx, ctx->croom.as<uint32_t>() + n_gb, n_gb, ctx->coff.as<uint64_t>()));
            const uint64_t max_vals = std::min<uint64_t>((uint64_t)n_used * CH, (uint64_t)gb * pl.table_size);
            RC_TRY(ensure(ctx, ctx->cvals, (max_vals / 2 + n_gb + 1) * 8 + 256));   // (+256: km_write_c_kernel reads up to 32 values from a row's start)
            RC_TRY(ensure(ctx, ctx->rank, (size_t)gb * (ctx->mask_words / 2) * 4));
            RC_TRY(ensure(ctx, ctx->vaddr, (size_t)gb * (ctx->mask_words / 2) * 8));
            RC_TRY(ensure(ctx, ctx->ucnt, n_gb * 4));
        }
        if (n_used) {
            // (grid-stride: without the host's count the grid covers the chunks the call expects, not the pool's capacity)
            const uint64_t n_grid = spec ? std::min<uint64_t>(n_used, expected_chunks * 2 + n_waves * NBK + 4096) : n_used;
            hipLaunchKernelGGL(chunk_scatter_kernel, dim3((uint32_t)std::min<uint64_t>((n_grid + 255) / 256, 1u << 16)), dim3(256), 0, ctx->stream,
                               ctx->desc.as<unsigned long long>(), n_used, ctx->gboff.as<uint64_t>(), d_gbcur,
                               ctx->clist.as<uint2>(), n_used_dev);
            HIP_TRY(ctx, hipGetLastError());
        }
        const int ev_c = sw.mark();
        RC_TRY(dispatch_stream_pass2(ctx, pl.sigma, pl.k, (uint32_t)n_gb, pl.table_size, ctx->table.as<uint32_t>(), s_compress));
        const int ev_d = sw.mark();
        if ((spec || n_ovf > 0) && !ovf_in_pool) RC_TRY(dispatch_overflow(ctx, pl.sigma, pl.k, p));
        if (s_compress) {
            ctx->table_compressed = true; ctx->comp_own_vals = true;
            ctx->comp_nb = NBK; ctx->comp_stride = 1; ctx->comp_tbl = stream_tbl_value(pl.sigma, pl.k);
        }
        ctx->mask_valid = true;
        ctx->main_kernel = use_quad ? "score_quad_kernel" : "score_stream_kernel";
        // the batch's counters: copied to pinned words, read at the caller's next wait (defer) or at this one
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rb, ctx->small, 64, hipMemcpyDeviceToHost, ctx->stream));
        ipkgpu_ctx::Pending& pd = ctx->pend;
        pd.sw.reset(sw_own.release());
        pd.ev_a = ev_a; pd.ev_b = ev_b; pd.ev_c = ev_c; pd.ev_d = ev_d;
        pd.cap = cap; pd.max_chunks = max_chunks; pd.windows = windows; pd.gb = gb; pd.pool_ovf_ok = pool_ovf_ok; pd.use_quad = use_quad;
        pd.active = true;
        if (defer) return IPKGPU_OK;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int rcf = stream_batch_check(ctx);
        if (rcf == IPKGPU_RETRY_AGAIN) { want = std::max<uint64_t>(want, ctx->pool_want_min); continue; }
        return rcf;
    }
    return fail(ctx, IPKGPU_ERR_NOMEM, "pair pool could not be sized");
}

// The part of score_batch that needs the batch's counters on the host; after a wait on the stream.  IPKGPU_RETRY_AGAIN /
// IPKGPU_RETRY_SMALLER: the pair pool ran out (found only now, in a call that skipped the wait after pass 1) -- score the batch again.
int score_batch_finish(ipkgpu_ctx* ctx, const Plan& pl, uint32_t g0, uint32_t gb, uint64_t* emitted_acc)
{
    if (ctx->pend.active) RC_TRY(stream_batch_check(ctx));
    unsigned long long e = ctx->emitted_host;
    if (!ctx->emitted_fetched) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_rb, ctx->small, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(&e, ctx->h_rb, 8);
    }
    *emitted_acc += e;
    // calibration of the next call's pair pool and workgroup count: PAIRS per window (not chunks -- chunk
    // counts include every wave's open chunks and would feed back into the workgroup count)
    uint64_t nb = 0;
    for (uint32_t i = 0; i < pl.n_mats; ++i) nb += (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb);
    if (nb) ctx->pairs_per_window = (double)e / ((double)nb * pl.nwin);
    return IPKGPU_OK;
}

// defer: the caller waits on the stream later anyway; if ctx->pend.active on return, it owes score_batch_finish after that wait
int score_batch(ipkgpu_ctx* ctx, const Plan& pl, const float* logp_dev, uint32_t g0, uint32_t gb,
                std::vector<uint32_t>& idx_host, uint64_t* emitted_acc, bool defer = false)
{
    ctx->emitted_fetched = false;
    ctx->pend.active = false; ctx->pend.sw.reset();
    RC_TRY(score_batch_impl(ctx, pl, logp_dev, g0, gb, idx_host, defer));
    if (ctx->pend.active) return IPKGPU_OK;
    return score_batch_finish(ctx, pl, g0, gb, emitted_acc);
}

// out[0..n] = exclusive scan of in[0..n)   (u32 -> u64; out[n] = the total): one single-pass device scan (rocPRIM, decoupled
// look-back) over n + 1 items, the last one a zero -- two launches where the three-kernel scan (block sums, one workgroup over
// the sums, apply) cost 27 us per call at a million keys, four times per scoring call.
// (the scans' inputs are read through global pointers: as members of a functor they are generic ones to the compiler, and their
//  loads FLAT -- tools/isa_flat.py)
#if defined(__HIP_DEVICE_COMPILE__)
#define IPK_GLOBAL_U32(p) ((ipkgpu::global_u32_ptr)(p))
#else
#define IPK_GLOBAL_U32(p) (p)
#endif
struct ScanU32In {
    const uint32_t* in; uint64_t n;
    __host__ __device__ uint64_t operator()(uint64_t i) const { return i < n ? (uint64_t)IPK_GLOBAL_U32(in)[i] : 0ull; }
};
struct ScanNonzeroIn {
    const uint32_t* in; uint64_t n;
    __host__ __device__ uint64_t operator()(uint64_t i) const { return i < n ? (uint64_t)(IPK_GLOBAL_U32(in)[i] != 0u) : 0ull; }
};
// out[0..n] = exclusive scan of (in[i] != 0): the position of every non-empty slot in the key list, out[n] = the number of keys
int scan_nonzero_u32(ipkgpu_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out)
{
    if (n == 0) { HIP_TRY(ctx, hipMemsetAsync(out, 0, 8, ctx->stream)); return IPKGPU_OK; }
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), ScanNonzeroIn{in, n});
    size_t bytes = 0;
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    RC_TRY(ensure(ctx, ctx->scan_sums, bytes));
    HIP_TRY(ctx, rocprim::exclusive_scan(ctx->scan_sums.p, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    return IPKGPU_OK;
}
struct ScanRowsIn {
    const uint32_t* const* rows; uint64_t slots, n;
    __host__ __device__ uint64_t operator()(uint64_t i) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const uint32_t* const __attribute__((address_space(1)))* rows_ptr;
        return i < n ? (uint64_t)IPK_GLOBAL_U32(((rows_ptr)rows)[i / slots])[i % slots] : 0ull;
#else
        return i < n ? (uint64_t)rows[i / slots][i % slots] : 0ull;
#endif
    }
};
// out[0 .. S * slots] = exclusive scan over S rows of `slots` counts laid end to end (rows: device array of device pointers)
int scan_rows_u32(ipkgpu_ctx* ctx, const uint32_t* const* rows, uint32_t S, uint64_t slots, uint64_t* out)
{
    const uint64_t n = (uint64_t)S * slots;
    if (n == 0) { HIP_TRY(ctx, hipMemsetAsync(out, 0, 8, ctx->stream)); return IPKGPU_OK; }
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), ScanRowsIn{rows, slots, n});
    size_t bytes = 0;
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    RC_TRY(ensure(ctx, ctx->scan_sums, bytes));
    HIP_TRY(ctx, rocprim::exclusive_scan(ctx->scan_sums.p, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    return IPKGPU_OK;
}
int scan_u32(ipkgpu_ctx* ctx, const uint32_t* in, uint64_t n, uint64_t* out)
{
    if (n == 0) { HIP_TRY(ctx, hipMemsetAsync(out, 0, 8, ctx->stream)); return IPKGPU_OK; }
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint64_t>(0), ScanU32In{in, n});
    size_t bytes = 0;
    HIP_TRY(ctx, rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    RC_TRY(ensure(ctx, ctx->scan_sums, bytes));
    HIP_TRY(ctx, rocprim::exclusive_scan(ctx->scan_sums.p, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), ctx->stream));
    return IPKGPU_OK;
}

}  // namespace

extern "C" {

// ---- group-major output ------------------------------------------------------------------------------
int ipkgpu_score_groups_device(ipkgpu_ctx* ctx, const float* logp_dev, uint32_t n_mats, uint32_t sites,
                               uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                               ipkgpu_result** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    Plan pl;
    RC_TRY(make_plan(ctx, logp_dev, n_mats, sites, sigma, mat_group, k, log_eps, pl));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_groups = pl.n_groups, cpg = pl.chunks_per_group;

    ipkgpu_result* res = new (std::nothrow) ipkgpu_result();
    if (!res) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    res->ctx = ctx;
    res->group_ids = pl.group_ids;
    res->offsets.assign((size_t)n_groups + 1, 0);
    struct Guard { ipkgpu_result* r; ~Guard() { if (r) ipkgpu_result_free(r); } } guard{res};

    RC_TRY(ensure(ctx, ctx->counts, (size_t)(pl.gpb * cpg) * 4));
    RC_TRY(ensure(ctx, ctx->offsets, (size_t)(pl.gpb * cpg + 1) * 8));
    RC_TRY(ensure(ctx, ctx->goff, (size_t)(pl.gpb + 1) * 8));

    ctx->acc_main_ms = ctx->acc_reduce_ms = ctx->acc_count_ms = ctx->acc_write_ms = ctx->acc_km_ms = 0;
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t_begin = sw.mark();
    RC_TRY(run_prefix(ctx, pl, logp_dev));
    const int t_pre = sw.mark();
    std::vector<std::pair<int, int>> ev_score, ev_compact;
    std::vector<uint32_t> idx_host;
    uint64_t total_entries = 0;
    uint32_t gpb_now = (uint32_t)pl.gpb, gb = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += gb) {
        gb = std::min<uint32_t>(gpb_now, n_groups - g0);
        const int s0 = sw.mark();
        const int rcb = score_batch(ctx, pl, logp_dev, g0, gb, idx_host, &res->emitted);
        if (rcb == IPKGPU_RETRY_SMALLER) { gpb_now = std::max<uint32_t>(1, gb / 2); gb = 0; continue; }   // the batch again, half as many groups
        if (rcb) return rcb;
        const int s1 = sw.mark();
        ev_score.push_back({s0, s1});
        res->score_launches += 1;

        const uint32_t n_chunks = ctx->table_compressed ? gb * ctx->comp_nb : gb * cpg;    // compressed form: one item per slice
        const uint32_t per_group = ctx->table_compressed ? ctx->comp_nb : cpg;
        if (ctx->table_compressed) {
            RC_TRY(ensure(ctx, ctx->offsets, ((size_t)n_chunks + 1) * 8));
            RC_TRY(scan_u32(ctx, ctx->ucnt.as<uint32_t>(), n_chunks, ctx->offsets.as<uint64_t>()));
            if (total_entries)
                hipLaunchKernelGGL(add_base_kernel, dim3((n_chunks + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                                   ctx->offsets.as<uint64_t>(), (uint64_t)n_chunks + 1, total_entries);
        } else {
            if (ctx->mask_valid)
                hipLaunchKernelGGL(count_chunks_mask_kernel, dim3(n_chunks), dim3(256), 0, ctx->stream,
                                   ctx->mask.as<uint32_t>(), ctx->mask_words, cpg, ctx->counts.as<uint32_t>());
            else
                hipLaunchKernelGGL(count_chunks_kernel, dim3(n_chunks), dim3(256), 0, ctx->stream,
                                   ctx->table.as<uint32_t>(), pl.table_size, cpg, ctx->counts.as<uint32_t>());
            hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                               ctx->counts.as<uint32_t>(), (uint64_t)n_chunks, total_entries, ctx->offsets.as<uint64_t>());
        }
        hipLaunchKernelGGL(gather_offsets_kernel, dim3((gb + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->offsets.as<uint64_t>(), per_group, gb + 1, ctx->goff.as<uint64_t>());
        HIP_TRY(ctx, hipGetLastError());
        std::vector<uint64_t> goff((size_t)gb + 1);
        HIP_TRY(ctx, hipMemcpyAsync(goff.data(), ctx->goff.p, ((size_t)gb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t g = 0; g <= gb; ++g) res->offsets[g0 + g] = goff[g];
        const uint64_t new_total = goff[gb];
        if (new_total > res->cap) {
            // grow the output (exact when the last batch is reached; doubling across batches)
            const size_t new_cap = (g0 + gb >= n_groups) ? (size_t)new_total : (size_t)std::max<uint64_t>(new_total, 2 * res->cap);
            uint32_t* nk = nullptr; float* ns = nullptr;
            HIP_TRY(ctx, ctx_alloc(ctx, (void**)&nk, std::max<size_t>(new_cap, 1) * 4));
            hipError_t e2 = ctx_alloc(ctx, (void**)&ns, std::max<size_t>(new_cap, 1) * 4);
            if (e2 != hipSuccess) { ctx_release(ctx, nk); HIP_TRY(ctx, e2); }
            if (total_entries) {
                (void)hipMemcpyAsync(nk, res->d_keys, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipMemcpyAsync(ns, res->d_scores, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipStreamSynchronize(ctx->stream);
            }
            ctx_release(ctx, res->d_keys);
            ctx_release(ctx, res->d_scores);
            res->d_keys = nk; res->d_scores = ns; res->cap = new_cap;
        }
        if (ctx->table_compressed) {
            const CompTable ct = comp_table(ctx);
            if (sigma == 4)
                hipLaunchKernelGGL(write_group_c_kernel<4>, dim3(n_chunks), dim3(256), 0, ctx->stream, ct,
                                   pl.table_size, (int)k, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores);
            else
                hipLaunchKernelGGL(write_group_c_kernel<20>, dim3(n_chunks), dim3(256), 0, ctx->stream, ct,
                                   pl.table_size, (int)k, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores);
        } else if (sigma == 4)
            hipLaunchKernelGGL(write_chunks_kernel<4>, dim3(n_chunks), dim3(256), 0, ctx->stream, ctx->table.as<uint32_t>(),
                               pl.table_size, cpg, (int)k, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores);
        else
            hipLaunchKernelGGL(write_chunks_kernel<20>, dim3(n_chunks), dim3(256), 0, ctx->stream, ctx->table.as<uint32_t>(),
                               pl.table_size, cpg, (int)k, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores);
        HIP_TRY(ctx, hipGetLastError());
        ev_compact.push_back({s1, sw.mark()});
        total_entries = new_total;
    }
    const int t_end = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    res->t_total = sw.ms(t_begin, t_end);
    res->t_prefix = sw.ms(t_begin, t_pre);
    for (auto& pr : ev_score) res->t_score += sw.ms(pr.first, pr.second);
    res->t_main = ctx->acc_main_ms; res->t_reduce = ctx->acc_reduce_ms;
    for (auto& pr : ev_compact) res->t_compact += sw.ms(pr.first, pr.second);
    guard.r = nullptr;
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
    if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(ctx, e); }
    const int rc = ipkgpu_score_groups_device(ctx, d, n_mats, sites, sigma, mat_group, k, log_eps, out);
    (void)hipFree(d);
    return rc;
}

// ---- KEEP_POSITIONS variant -------------------------------------------
int ipkgpu_score_groups_positions(ipkgpu_ctx* ctx, const float* logp, uint32_t n_mats, uint32_t sites, uint32_t sigma,
                                  const uint32_t* mat_group, uint32_t k, float log_eps, ipkgpu_result** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    Plan pl;
    RC_TRY(make_plan(ctx, logp, n_mats, sites, sigma, mat_group, k, log_eps, pl));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((uint64_t)n_mats * pl.nwin >= 0xFFFFFFFFull) return fail(ctx, IPKGPU_ERR_INVALID, "too many windows per group for the position code");
    pl.gpb = std::max<uint64_t>(1, pl.gpb / 2);                      // 8-byte table entries
    const uint32_t n_groups = pl.n_groups, cpg = pl.chunks_per_group;

    ipkgpu_result* res = new (std::nothrow) ipkgpu_result();
    if (!res) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    res->ctx = ctx;
    res->group_ids = pl.group_ids;
    res->offsets.assign((size_t)n_groups + 1, 0);
    struct Guard { ipkgpu_result* r; ~Guard() { if (r) ipkgpu_result_free(r); } } guard{res};

    // host matrices in; rank of every matrix inside its group = processing order of explore_group (:641)
    const size_t bytes = (size_t)n_mats * sites * sigma * 4;
    float* d_logp = nullptr;
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&d_logp, std::max<size_t>(bytes, 4)));
    struct LGuard { ipkgpu_ctx* c; void* p; ~LGuard() { ctx_release(c, p); } } lguard{ctx, d_logp};
    HIP_TRY(ctx, hipMemcpy(d_logp, logp, bytes, hipMemcpyHostToDevice));
    std::vector<uint32_t> rank(n_mats), seen(n_groups, 0);
    for (uint32_t i = 0; i < n_mats; ++i) rank[i] = seen[pl.slot_of[i]]++;
    RC_TRY(ensure(ctx, ctx->branch, (size_t)n_mats * 4));
    HIP_TRY(ctx, hipMemcpy(ctx->branch.p, rank.data(), (size_t)n_mats * 4, hipMemcpyHostToDevice));

    RC_TRY(ensure(ctx, ctx->counts, (size_t)(pl.gpb * cpg) * 4));
    RC_TRY(ensure(ctx, ctx->offsets, (size_t)(pl.gpb * cpg + 1) * 8));
    RC_TRY(ensure(ctx, ctx->goff, (size_t)(pl.gpb + 1) * 8));
    ctx->acc_main_ms = ctx->acc_reduce_ms = ctx->acc_count_ms = ctx->acc_write_ms = ctx->acc_km_ms = 0;
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t_begin = sw.mark();
    RC_TRY(run_prefix(ctx, pl, d_logp));
    std::vector<uint32_t> idx_host((size_t)n_mats * 2);
    uint64_t total_entries = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += (uint32_t)pl.gpb) {
        const uint32_t gb = std::min<uint32_t>((uint32_t)pl.gpb, n_groups - g0);
        uint32_t nb = 0;
        for (uint32_t i = 0; i < n_mats; ++i) {
            idx_host[n_mats + i] = 0;
            if (pl.slot_of[i] >= g0 && pl.slot_of[i] < g0 + gb) { idx_host[nb++] = i; idx_host[n_mats + i] = pl.slot_of[i] - g0; }
        }
        RC_TRY(ensure(ctx, ctx->idx, (size_t)n_mats * 8));
        RC_TRY(ensure(ctx, ctx->table, (size_t)gb * pl.table_size * 8));
        RC_TRY(ensure(ctx, ctx->ovfq, (size_t)nb * pl.nwin * 8));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->idx.p, idx_host.data(), (size_t)n_mats * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ScoreParams p;
        p.logp = d_logp; p.best = ctx->best.as<float>();
        p.mat_list = ctx->idx.as<uint32_t>(); p.mat_slot = ctx->idx.as<uint32_t>() + n_mats;
        p.n_batch_mats = nb; p.sites = sites; p.nwin = pl.nwin; p.tiles_per_mat = pl.tiles_per_mat; p.eps = log_eps;
        p.table = ctx->table.p; p.table_size = pl.table_size; p.mat_rank = ctx->branch.as<uint32_t>();
        p.mask = nullptr; p.mask_words = 0;
        p.emitted = reinterpret_cast<unsigned long long*>(ctx->small);
        p.ovf_queue = ctx->ovfq.as<unsigned long long>();
        p.ovf_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 16);
        p.flags = 0;
        HIP_TRY(ctx, hipMemsetAsync(ctx->table.p, 0, (size_t)gb * pl.table_size * 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(p.ovf_count, 0, 4, ctx->stream));
        RC_TRY(dispatch_score_pos(ctx, sigma, k, p));
        res->score_launches += 1;
        const uint32_t n_chunks = gb * cpg;
        hipLaunchKernelGGL(count_chunks64_kernel, dim3(n_chunks), dim3(256), 0, ctx->stream,
                           ctx->table.as<unsigned long long>(), pl.table_size, cpg, ctx->counts.as<uint32_t>());
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           ctx->counts.as<uint32_t>(), (uint64_t)n_chunks, total_entries, ctx->offsets.as<uint64_t>());
        hipLaunchKernelGGL(gather_offsets_kernel, dim3((gb + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->offsets.as<uint64_t>(), cpg, gb + 1, ctx->goff.as<uint64_t>());
        HIP_TRY(ctx, hipGetLastError());
        std::vector<uint64_t> goff((size_t)gb + 1);
        HIP_TRY(ctx, hipMemcpyAsync(goff.data(), ctx->goff.p, ((size_t)gb + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t g = 0; g <= gb; ++g) res->offsets[g0 + g] = goff[g];
        const uint64_t new_total = goff[gb];
        if (new_total > res->cap) {
            const size_t new_cap = (g0 + gb >= n_groups) ? (size_t)new_total : (size_t)std::max<uint64_t>(new_total, 2 * res->cap);
            uint32_t *nk = nullptr, *np = nullptr; float* ns = nullptr;
            HIP_TRY(ctx, ctx_alloc(ctx, (void**)&nk, std::max<size_t>(new_cap, 1) * 4));
            hipError_t e2 = ctx_alloc(ctx, (void**)&ns, std::max<size_t>(new_cap, 1) * 4);
            hipError_t e3 = e2 == hipSuccess ? ctx_alloc(ctx, (void**)&np, std::max<size_t>(new_cap, 1) * 4) : e2;
            if (e3 != hipSuccess) { ctx_release(ctx, nk); ctx_release(ctx, ns); HIP_TRY(ctx, e3); }
            if (total_entries) {
                (void)hipMemcpyAsync(nk, res->d_keys, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipMemcpyAsync(ns, res->d_scores, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipMemcpyAsync(np, res->d_positions, total_entries * 4, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipStreamSynchronize(ctx->stream);
            }
            ctx_release(ctx, res->d_keys); ctx_release(ctx, res->d_scores); ctx_release(ctx, res->d_positions);
            res->d_keys = nk; res->d_scores = ns; res->d_positions = np; res->cap = new_cap;
        }
        if (sigma == 4)
            hipLaunchKernelGGL(write_chunks_pos_kernel<4>, dim3(n_chunks), dim3(256), 0, ctx->stream, ctx->table.as<unsigned long long>(),
                               pl.table_size, cpg, (int)k, pl.nwin, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores, res->d_positions);
        else
            hipLaunchKernelGGL(write_chunks_pos_kernel<20>, dim3(n_chunks), dim3(256), 0, ctx->stream, ctx->table.as<unsigned long long>(),
                               pl.table_size, cpg, (int)k, pl.nwin, ctx->offsets.as<uint64_t>(), res->d_keys, res->d_scores, res->d_positions);
        HIP_TRY(ctx, hipGetLastError());
        total_entries = new_total;
    }
    const int t_end = sw.mark();
    unsigned long long emitted = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&emitted, ctx->small, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    res->emitted = emitted;
    res->t_total = sw.ms(t_begin, t_end);
    guard.r = nullptr;
    *out = res;
    return IPKGPU_OK;
}

const uint32_t* ipkgpu_result_positions(ipkgpu_result* r)
{
    if (!r || !r->d_positions) return nullptr;
    if (!r->h_positions_ok) {
        const size_t n = (size_t)r->offsets.back();
        r->h_positions.resize(std::max<size_t>(n, 1));
        (void)hipSetDevice(r->ctx->device);
        if (n && hipMemcpy(r->h_positions.data(), r->d_positions, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
        r->h_positions_ok = true;
    }
    return r->h_positions.data();
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
        (void)hipSetDevice(r->ctx->device);
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
        (void)hipSetDevice(r->ctx->device);
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
        case IPKGPU_T_SCORE_MAIN: return r->t_main;
        case IPKGPU_T_SCORE_REDUCE: return r->t_reduce;
    }
    return 0;
}

void ipkgpu_result_free(ipkgpu_result* r)
{
    if (!r) return;
    if (r->ctx) { (void)hipSetDevice(r->ctx->device); ctx_release(r->ctx, r->d_keys); ctx_release(r->ctx, r->d_scores); ctx_release(r->ctx, r->d_positions); }
    delete r;
}

}  // extern "C"

// ---- key-major output: database parts and their merge ----------------------------------------------
namespace {

// Merges S sources of one owner: source s brings its counts row counts_rows[s] ([slots], device) and its entry block
// src_rows[s] (device).  Writes dst entries (n_total), dst_off [slots+1] into ctx->tmp_b, and, when `db` is given, the
// compact key list.
int merge_sources(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t P, uint32_t S, uint64_t slots,
                  const std::vector<const uint32_t*>& counts_rows, const std::vector<const uint2*>& src_rows,
                  uint32_t* total_out /*[slots] device*/, uint2** dst_out, uint64_t* n_total_out, ipkgpu_db* db)
{
    // workspaces: tmp_a = flags u32[slots] ; tmp_b = dst_off u64[slots+1] ; tmp_c = ONE scan over the sources' counts rows laid end to
    // end, u64[S * slots + 1] (it was a scan per source: 2 S launches) ; ptrs = the pointer arrays (uploaded from pinned staging: no wait)
    RC_TRY(ensure(ctx, ctx->tmp_a, slots * 4));
    RC_TRY(ensure(ctx, ctx->tmp_b, (slots + 1) * 8));
    RC_TRY(ensure(ctx, ctx->tmp_c, ((size_t)S * slots + 1) * 8));
    RC_TRY(ensure(ctx, ctx->ptrs, (size_t)S * 16));
    void* stage = nullptr;
    RC_TRY(upload_stage(ctx, (size_t)S * 16, &stage));
    const void** hp = reinterpret_cast<const void**>(stage);
    for (uint32_t s = 0; s < S; ++s) { hp[s] = counts_rows[s]; hp[S + s] = src_rows[s]; }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ptrs.p, hp, (size_t)S * 16, hipMemcpyHostToDevice, ctx->stream));
    RC_TRY(upload_staged(ctx));
    const uint32_t* const* d_counts = ctx->ptrs.as<const uint32_t*>();
    const uint2* const* d_src = reinterpret_cast<const uint2* const*>(ctx->ptrs.as<const void*>() + S);
    const uint32_t nb256 = (uint32_t)((slots + 255) / 256);
    hipLaunchKernelGGL(merge_sum_counts_kernel, dim3(nb256), dim3(256), 0, ctx->stream, d_counts, S, slots, total_out,
                       ctx->tmp_a.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    RC_TRY(scan_u32(ctx, total_out, slots, ctx->tmp_b.as<uint64_t>()));
    RC_TRY(scan_rows_u32(ctx, d_counts, S, slots, ctx->tmp_c.as<uint64_t>()));
    uint64_t* h_tot = reinterpret_cast<uint64_t*>(ctx->h_rb) + RB_OWNER_OFF;        // (pinned: [0] entries, [1] keys)
    HIP_TRY(ctx, hipMemcpyAsync(h_tot, ctx->tmp_b.as<uint64_t>() + slots, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n_total = h_tot[0];
    uint2* dst = nullptr;
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&dst, std::max<uint64_t>(n_total, 1) * 8));
    for (uint64_t first = 0; first < slots; first += WAVE_PER_ITEM_SPAN) {   // (a wavefront per slot: 4^13 slots in one launch would be 2^32 threads)
        const uint64_t span = std::min<uint64_t>(WAVE_PER_ITEM_SPAN, slots - first);
        hipLaunchKernelGGL(merge_copy_kernel, dim3((uint32_t)((span + 3) / 4)), dim3(256), 0, ctx->stream, d_counts, S, slots,
                           ctx->tmp_c.as<uint64_t>(), d_src, ctx->tmp_b.as<uint64_t>(), dst, first);
    }
    *dst_out = dst;                       // owned by the caller from here on (also on failure)
    *n_total_out = n_total;
    HIP_TRY(ctx, hipGetLastError());
    if (db) {
        // compact key list: scan of flags; the key arrays are sized for min(slots, entries) keys, the count comes back with the
        // caller's next wait on the stream (db->n_keys is read by the callers only after theirs)
        RC_TRY(ensure(ctx, ctx->offsets, (slots + 1) * 8));
        RC_TRY(scan_u32(ctx, ctx->tmp_a.as<uint32_t>(), slots, ctx->offsets.as<uint64_t>()));
        HIP_TRY(ctx, hipMemcpyAsync(h_tot + 1, ctx->offsets.as<uint64_t>() + slots, 8, hipMemcpyDeviceToHost, ctx->stream));
        const uint64_t keys_bound = std::max<uint64_t>(1, std::min<uint64_t>(slots, n_total));
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_keys, keys_bound * 4));
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_key_off, (keys_bound + 1) * 8));
        const uint32_t nbk = (uint32_t)((slots + 1 + 255) / 256);
        if (sigma == 4)
            hipLaunchKernelGGL(merge_write_keys_kernel<4>, dim3(nbk), dim3(256), 0, ctx->stream, total_out, ctx->offsets.as<uint64_t>(),
                               ctx->tmp_b.as<uint64_t>(), slots, owner, P, (int)k, db->d_keys, db->d_key_off);
        else
            hipLaunchKernelGGL(merge_write_keys_kernel<20>, dim3(nbk), dim3(256), 0, ctx->stream, total_out, ctx->offsets.as<uint64_t>(),
                               ctx->tmp_b.as<uint64_t>(), slots, owner, P, (int)k, db->d_keys, db->d_key_off);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                 // (the key count; also the end of the merge for the callers' timing)
        const uint64_t n_keys = h_tot[1];
        db->n_keys = n_keys;
    }
    return IPKGPU_OK;
}

}  // namespace

extern "C" {

int ipkgpu_score_groups_keymajor_device(ipkgpu_ctx* ctx, const float* logp_dev, uint32_t n_mats, uint32_t sites,
                                        uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                                        uint32_t n_owners, ipkgpu_parts** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (n_owners == 0) return fail(ctx, IPKGPU_ERR_INVALID, "n_owners must be >= 1");
    if (n_mats == 0) {
        // a rank without branch groups (more ranks than groups): empty parts, so that it still takes part in the exchange
        if ((sigma != 4 && sigma != 20) || k < 2 || k > ipkgpu_max_k(sigma)) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        ipkgpu_parts* e = new (std::nothrow) ipkgpu_parts();
        if (!e) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
        e->ctx = ctx; e->n_owners = n_owners; e->slots = (ipow(sigma, (int)k) + n_owners - 1) / n_owners;
        e->owner_off.assign((size_t)n_owners + 1, 0);
        struct EGuard { ipkgpu_parts* r; ~EGuard() { if (r) ipkgpu_parts_free(r); } } eg{e};
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&e->d_counts, (size_t)n_owners * e->slots * 4));
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&e->d_entries, 8));
        HIP_TRY(ctx, hipMemsetAsync(e->d_counts, 0, (size_t)n_owners * e->slots * 4, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        eg.r = nullptr;
        *out = e;
        return IPKGPU_OK;
    }
    Plan pl;
    RC_TRY(make_plan(ctx, logp_dev, n_mats, sites, sigma, mat_group, k, log_eps, pl));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t T = pl.table_size;
    const uint32_t P = n_owners;
    const uint64_t slots = (T + P - 1) / P;
    const uint32_t n_groups = pl.n_groups;

    ipkgpu_parts* parts = new (std::nothrow) ipkgpu_parts();
    if (!parts) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    parts->ctx = ctx; parts->n_owners = P; parts->slots = slots;
    parts->owner_off.assign((size_t)P + 1, 0);
    struct Guard { ipkgpu_parts* r; ~Guard() { if (r) ipkgpu_parts_free(r); } } guard{parts};

    RC_TRY(ensure(ctx, ctx->branch, (size_t)n_groups * 4));
    ctx->h_branch = pl.group_ids;                          // (context-owned copy: no wait for the upload, whatever path this call leaves by)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->branch.p, ctx->h_branch.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, ctx->stream));

    ctx->acc_main_ms = ctx->acc_reduce_ms = ctx->acc_count_ms = ctx->acc_write_ms = ctx->acc_km_ms = 0;
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t_begin = sw.mark();
    RC_TRY(run_prefix(ctx, pl, logp_dev));
    const int t_pre = sw.mark();
    std::vector<std::pair<int, int>> ev_score, ev_compact, ev_km;
    std::pair<int, int> ev_keys{-1, -1};
    std::vector<uint32_t> idx_host;

    struct Batch { uint32_t* counts = nullptr; uint2* entries = nullptr; std::vector<uint64_t> owner_off; };
    std::vector<Batch> batches;
    auto free_batches = [&]() { for (auto& b : batches) { ctx_release(ctx, b.counts); ctx_release(ctx, b.entries); } batches.clear(); };
    struct BGuard { decltype(free_batches)& f; ~BGuard() { f(); } } bguard{free_batches};

    const uint64_t n_slots_all = (uint64_t)P * slots;
    uint32_t gpb_now = (uint32_t)pl.gpb, gb = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += gb) {
        gb = std::min<uint32_t>(gpb_now, n_groups - g0);
        const int s0 = sw.mark();
        // (defer: a stream-variant batch returns without a wait of its own; what it owes -- the pool's state, the scored count --
        //  is settled at this loop's one wait, score_batch_finish below)
        const int rcb = score_batch(ctx, pl, logp_dev, g0, gb, idx_host, &parts->emitted, true);
        if (rcb == IPKGPU_RETRY_SMALLER) { gpb_now = std::max<uint32_t>(1, gb / 2); gb = 0; continue; }   // the batch again, half as many groups
        if (rcb) return rcb;
        const int s1 = sw.mark();
        ev_score.push_back({s0, s1});
        parts->score_launches += 1;

        batches.emplace_back();
        Batch& b = batches.back();
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&b.counts, n_slots_all * 4));
        HIP_TRY(ctx, hipMemsetAsync(b.counts, 0, n_slots_all * 4, ctx->stream));
        // compressed tables go through the fast key-major writer, up to 256 groups at a time: it wants those rows counted per quarter
        const uint32_t kmc_pass = (uint32_t)std::min<int64_t>(256, std::max<int64_t>(4, ctx->opt_kmc_pass > 0 ? ctx->opt_kmc_pass : IPK_KMC_PASS));
        const bool fast_c = ctx->table_compressed && 64ull * (ctx->mask_words / 2) * 8 < (1ull << 32);
        uint32_t* qpack = nullptr;
        if (fast_c) {
            RC_TRY(ensure(ctx, ctx->qpack, n_slots_all * 4));
            if (gb <= kmc_pass) qpack = ctx->qpack.as<uint32_t>();     // one pass: the total counts are the pass' counts
        }
        if (ctx->mask_valid && (T + 31) / 32 >= (uint64_t)ctx->num_cu * 1024)      // a thread per mask word fills the chip
            hipLaunchKernelGGL(km_count_mask_kernel, dim3((uint32_t)(((T + 31) / 32 + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->mask.as<uint32_t>(), ctx->mask_words, T, gb, P, slots, b.counts, qpack);
        else if (ctx->mask_valid)
            hipLaunchKernelGGL(km_count_mask_key_kernel, dim3((uint32_t)((T + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->mask.as<uint32_t>(), ctx->mask_words, T, gb, P, slots, b.counts, qpack);
        else
            hipLaunchKernelGGL(km_count_kernel, dim3((uint32_t)((T + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->table.as<uint32_t>(), T, gb, P, slots, b.counts);
        HIP_TRY(ctx, hipGetLastError());
        RC_TRY(ensure(ctx, ctx->offsets, (n_slots_all + 1) * 8));
        RC_TRY(scan_u32(ctx, b.counts, n_slots_all, ctx->offsets.as<uint64_t>()));
        // owner offsets of this batch
        RC_TRY(ensure(ctx, ctx->goff, ((size_t)P + 1) * 8));
        hipLaunchKernelGGL(gather_offsets_kernel, dim3((P + 1 + 255) / 256), dim3(256), 0, ctx->stream,
                           ctx->offsets.as<uint64_t>(), (uint32_t)slots, P + 1, ctx->goff.as<uint64_t>());
        HIP_TRY(ctx, hipGetLastError());
        b.owner_off.resize((size_t)P + 1);
        const bool rb_ok = P <= RB_OWNERS_MAX;                 // (the offsets go to pinned words: a copy the stream does not stall on)
        uint64_t* h_off = rb_ok ? reinterpret_cast<uint64_t*>(ctx->h_rb) + RB_OWNER_OFF : b.owner_off.data();
        HIP_TRY(ctx, hipMemcpyAsync(h_off, ctx->goff.p, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));

        // One owner, the whole call in one batch, a key space of moderate size: the database's key list is built here, from the counts
        // and their scan BEFORE the writer advances it -- ipkgpu_db_from_parts would scan the counts again, in a call of its own
        // with two more waits (a tenth of a 125-group share of cfg2; 1.3 ms at cfg4).  The arrays are sized for every slot (12 bytes each).
        // (larger key spaces -- AA k=6: 64 M slots -- only where the previous call's entries dwarf those 12 bytes per slot)
        const bool pre_keys = P == 1 && g0 == 0 && gb == n_groups && rb_ok && !(ctx->opt_flags & 64) &&
                              (slots <= (1ull << 24) || ctx->last_entries * 8 >= slots * 12 * 4);
        uint32_t* k_keys = nullptr; uint64_t* k_off = nullptr;
        struct KGuard { ipkgpu_ctx* c; uint32_t*& a; uint64_t*& b; ~KGuard() { ctx_release(c, a); ctx_release(c, b); } } kguard{ctx, k_keys, k_off};
        int kt0 = -1, kt1 = -1;
        if (pre_keys) {
            kt0 = sw.mark();
            RC_TRY(ensure(ctx, ctx->tmp_b, (slots + 1) * 8));
            RC_TRY(scan_nonzero_u32(ctx, b.counts, slots, ctx->tmp_b.as<uint64_t>()));
            HIP_TRY(ctx, hipMemcpyAsync(h_off + P + 1, ctx->tmp_b.as<uint64_t>() + slots, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, ctx_alloc(ctx, (void**)&k_keys, slots * 4));
            HIP_TRY(ctx, ctx_alloc(ctx, (void**)&k_off, (slots + 1) * 8));
            const uint32_t nbk = (uint32_t)((slots + 1 + 255) / 256);
            if (sigma == 4)
                hipLaunchKernelGGL(merge_write_keys_kernel<4>, dim3(nbk), dim3(256), 0, ctx->stream, b.counts, ctx->tmp_b.as<uint64_t>(),
                                   ctx->offsets.as<uint64_t>(), slots, 0u, 1u, (int)k, k_keys, k_off);
            else
                hipLaunchKernelGGL(merge_write_keys_kernel<20>, dim3(nbk), dim3(256), 0, ctx->stream, b.counts, ctx->tmp_b.as<uint64_t>(),
                                   ctx->offsets.as<uint64_t>(), slots, 0u, 1u, (int)k, k_keys, k_off);
            HIP_TRY(ctx, hipGetLastError());
            kt1 = sw.mark();
        }

        // the key-major writer over `cap_e` entries of room
        auto launch_writer = [&](uint64_t cap_e) -> int {
            if (ctx->table_compressed) {
                uint64_t per_xcd = (((T + 63) / 64) + 7) / 8;
                // (the writer that walks runs of consecutive key blocks; debug_flags bit 12: one workgroup per key block)
                // Measured (r04): 8 % faster at a cfg3 share (125 groups: 32 rows per wavefront), 9 % SLOWER at cfg4 and cfg5's passes (250 /
                // 256 groups: 64 rows per wavefront, where the writer already moves its bytes at 4.4 TB/s) -- so only up to 128 groups.
                const uint32_t rows_now = qpack ? gb : std::min<uint32_t>(gb, kmc_pass);
                const bool runs = !(ctx->opt_flags & 4096) && IPK_KMC_RUNS_DEFAULT && (rows_now <= 128 || (ctx->opt_flags & 8192));
                if (runs && (qpack || fast_c)) per_xcd = (kmc_runs(T, ctx->comp_tbl) + 7) / 8;
                if (qpack && runs)
                    KM_LAUNCH(km_write_c_run_kernel, KMC_CAP, comp_table(ctx), T, gb, ctx->branch.as<uint32_t>() + g0, P, slots, b.counts,
                              qpack, ctx->offsets.as<uint64_t>(), b.entries, cap_e);
                else if (qpack)
                    KM_LAUNCH(km_write_c_kernel, KMC_CAP, comp_table(ctx), T, gb, ctx->branch.as<uint32_t>() + g0, P, slots, b.counts,
                              qpack, ctx->offsets.as<uint64_t>(), b.entries, cap_e);
                else if (fast_c) {
                    // more than 256 groups: passes of 256, each with its own counts; the cursors (the scan of the TOTAL counts) advance
                    // from pass to pass, so a key's entries stay in group order
                    RC_TRY(ensure(ctx, ctx->pcounts, n_slots_all * 4));
                    for (uint32_t p0 = 0; p0 < gb; p0 += kmc_pass) {
                        const uint32_t pg = std::min<uint32_t>(kmc_pass, gb - p0);
                        const uint32_t* pmask = ctx->mask.as<uint32_t>() + (size_t)p0 * ctx->mask_words;
                        if ((T + 31) / 32 >= (uint64_t)ctx->num_cu * 1024)
                            hipLaunchKernelGGL(km_count_mask_kernel, dim3((uint32_t)(((T + 31) / 32 + 255) / 256)), dim3(256), 0, ctx->stream,
                                               pmask, ctx->mask_words, T, pg, P, slots, ctx->pcounts.as<uint32_t>(), ctx->qpack.as<uint32_t>());
                        else
                            hipLaunchKernelGGL(km_count_mask_key_kernel, dim3((uint32_t)((T + 255) / 256)), dim3(256), 0, ctx->stream,
                                               pmask, ctx->mask_words, T, pg, P, slots, ctx->pcounts.as<uint32_t>(), ctx->qpack.as<uint32_t>());
                        HIP_TRY(ctx, hipGetLastError());
                        CompTable ctp = comp_table(ctx);
                        ctp.mask = pmask;
                        ctp.vaddr += (size_t)p0 * (ctx->mask_words / 2);
                        ctp.rank += (size_t)p0 * (ctx->mask_words / 2);
                        if (runs)
                            KM_LAUNCH(km_write_c_run_kernel, KMC_CAP, ctp, T, pg, ctx->branch.as<uint32_t>() + g0 + p0, P, slots, ctx->pcounts.as<uint32_t>(),
                                      ctx->qpack.as<uint32_t>(), ctx->offsets.as<uint64_t>(), b.entries, cap_e);
                        else
                            KM_LAUNCH(km_write_c_kernel, KMC_CAP, ctp, T, pg, ctx->branch.as<uint32_t>() + g0 + p0, P, slots, ctx->pcounts.as<uint32_t>(),
                                      ctx->qpack.as<uint32_t>(), ctx->offsets.as<uint64_t>(), b.entries, cap_e);
                        HIP_TRY(ctx, hipGetLastError());
                    }
                } else
                    hipLaunchKernelGGL(km_write_c_generic_kernel, dim3((uint32_t)((((T + 63) / 64 + 7) / 8) * 8)), dim3(256), 0, ctx->stream,
                                       comp_table(ctx), T, gb, ctx->branch.as<uint32_t>() + g0, P, slots,
                                       ctx->offsets.as<uint64_t>(), b.entries, cap_e);
            } else if ((ctx->opt_flags & 512) || (gb < 96 && !(ctx->opt_flags & 1024)))
                // (the writer whose stores follow the tiles, not the lines: for fewer than ~96 groups a key has too few entries for the
                //  line-cut stores to repay their LDS pass -- 0.155 against 0.127 ms at 2 groups; debug_flags bit 9 forces it, bit 10 the other)
                hipLaunchKernelGGL(km_write_kernel, dim3((uint32_t)((T + 63) / 64)), dim3(256), 0, ctx->stream,
                                   ctx->table.as<uint32_t>(), T, gb, ctx->branch.as<uint32_t>() + g0, P, slots,
                                   ctx->offsets.as<uint64_t>(), b.entries, cap_e);
            else
                hipLaunchKernelGGL(km_write_lines_kernel, dim3((uint32_t)((T + 63) / 64)), dim3(256), 0, ctx->stream,
                                   ctx->table.as<uint32_t>(), T, gb, ctx->branch.as<uint32_t>() + g0, P, slots,
                                   ctx->offsets.as<uint64_t>(), b.entries, cap_e);
            HIP_TRY(ctx, hipGetLastError());
            return IPKGPU_OK;
        };
        // With the previous call's entries per group as the estimate, the writer's output is allocated and the writer launched
        // BEFORE the host knows the total: the batch's one wait then covers scoring, counting and writing.  A writer that finds
        // less room than the total leaves everything untouched (kernels_keymajor.hpp) and runs again below, after that wait.
        uint64_t cap_e = 0;
        const bool spec_e = rb_ok && ctx->last_gb > 0 && ctx->last_entries > 0 && !(ctx->opt_flags & 64);
        int km0 = -1, km1 = -1;
        if (spec_e) {
            const uint64_t est = (uint64_t)((unsigned __int128)ctx->last_entries * gb / ctx->last_gb);
            size_t got = 0;
            if (ctx_alloc_atleast(ctx, (void**)&b.entries, est * 8, (est + est / 20 + 4096) * 8, &got) == hipSuccess) cap_e = got / 8;
            else { (void)hipGetLastError(); b.entries = nullptr; }
        }
        if (cap_e) { km0 = sw.mark(); RC_TRY(launch_writer(cap_e)); km1 = sw.mark(); }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));              // the batch's one wait
        const int rcf = ctx->pend.active ? score_batch_finish(ctx, pl, g0, gb, &parts->emitted) : IPKGPU_OK;
        if (rcf == IPKGPU_RETRY_SMALLER || rcf == IPKGPU_RETRY_AGAIN) {
            // the pair pool ran out (a call that skipped the wait after pass 1 learns it only here): the batch again
            ctx_release(ctx, b.counts); ctx_release(ctx, b.entries);
            batches.pop_back();
            ev_score.pop_back(); parts->score_launches -= 1;
            if (rcf == IPKGPU_RETRY_SMALLER) gpb_now = std::max<uint32_t>(1, gb / 2);
            gb = 0;
            continue;
        }
        if (rcf) return rcf;
        if (rb_ok) memcpy(b.owner_off.data(), h_off, ((size_t)P + 1) * 8);
        if (b.owner_off[P] > cap_e || cap_e == 0) {
            ctx_release(ctx, b.entries); b.entries = nullptr;
            HIP_TRY(ctx, ctx_alloc(ctx, (void**)&b.entries, std::max<uint64_t>(b.owner_off[P], 1) * 8));
            km0 = sw.mark(); RC_TRY(launch_writer(std::max<uint64_t>(b.owner_off[P], 1))); km1 = sw.mark();
        }
        ctx->last_entries = b.owner_off[P]; ctx->last_gb = gb;
        if (pre_keys) {
            parts->pre_keys = k_keys; parts->pre_key_off = k_off; parts->pre_n_keys = h_off[P + 1];
            k_keys = nullptr; k_off = nullptr;
            ev_keys = {kt0, kt1};
        }
        ev_km.push_back({km0, km1});
        ev_compact.push_back({s1, km1});
    }

    if (batches.size() == 1) {
        parts->d_counts = batches[0].counts; parts->d_entries = batches[0].entries;
        parts->owner_off = batches[0].owner_off;
        batches[0].counts = nullptr; batches[0].entries = nullptr;
    } else {
        // several batches of groups: per owner, merge the batches (sources in batch = group order)
        const int m0 = sw.mark();
        const uint32_t S = (uint32_t)batches.size();
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&parts->d_counts, n_slots_all * 4));
        uint64_t grand = 0;
        for (auto& b : batches) grand += b.owner_off[P];
        HIP_TRY(ctx, ctx_alloc(ctx, (void**)&parts->d_entries, std::max<uint64_t>(grand, 1) * 8));
        uint64_t done = 0;
        int rc = IPKGPU_OK;
        for (uint32_t o = 0; o < P && rc == IPKGPU_OK; ++o) {
            std::vector<const uint32_t*> crow(S);
            std::vector<const uint2*> srow(S);
            for (uint32_t s = 0; s < S; ++s) { crow[s] = batches[s].counts + (size_t)o * slots; srow[s] = batches[s].entries + batches[s].owner_off[o]; }
            uint2* dst = nullptr; uint64_t n_total = 0;
            rc = merge_sources(ctx, sigma, k, o, P, S, slots, crow, srow, parts->d_counts + (size_t)o * slots, &dst, &n_total, nullptr);
            if (rc == IPKGPU_OK) {
                if (n_total) (void)hipMemcpyAsync(parts->d_entries + done, dst, n_total * 8, hipMemcpyDeviceToDevice, ctx->stream);
                (void)hipStreamSynchronize(ctx->stream);
                parts->owner_off[o] = done; done += n_total; parts->owner_off[o + 1] = done;
            }
            ctx_release(ctx, dst);
        }
        if (rc) return rc;
        ev_compact.push_back({m0, sw.mark()});
    }
    const int t_end = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    parts->t_total = sw.ms(t_begin, t_end);
    parts->t_prefix = sw.ms(t_begin, t_pre);
    for (auto& pr : ev_score) parts->t_score += sw.ms(pr.first, pr.second);
    for (auto& pr : ev_km) ctx->acc_km_ms += sw.ms(pr.first, pr.second);
    parts->t_main = ctx->acc_main_ms; parts->t_reduce = ctx->acc_reduce_ms;
    parts->t_count = ctx->acc_count_ms; parts->t_write = ctx->acc_write_ms; parts->t_km = ctx->acc_km_ms;
    for (auto& pr : ev_compact) parts->t_compact += sw.ms(pr.first, pr.second);
    parts->t_keys = sw.ms(ev_keys.first, ev_keys.second);
    guard.r = nullptr;
    *out = parts;
    return IPKGPU_OK;
}

uint32_t ipkgpu_parts_num_owners(const ipkgpu_parts* p) { return p ? p->n_owners : 0; }
uint64_t ipkgpu_parts_slots(const ipkgpu_parts* p) { return p ? p->slots : 0; }
const uint32_t* ipkgpu_parts_counts_device(const ipkgpu_parts* p) { return p ? p->d_counts : nullptr; }
const void* ipkgpu_parts_entries_device(const ipkgpu_parts* p) { return p ? p->d_entries : nullptr; }
const uint64_t* ipkgpu_parts_owner_offsets(const ipkgpu_parts* p) { return p ? p->owner_off.data() : nullptr; }
uint64_t ipkgpu_parts_emitted(const ipkgpu_parts* p) { return p ? p->emitted : 0; }
double ipkgpu_parts_time_ms(const ipkgpu_parts* p, int which)
{
    if (!p) return 0;
    switch (which) {
        case IPKGPU_T_TOTAL: return p->t_total;
        case IPKGPU_T_PREFIX: return p->t_prefix;
        case IPKGPU_T_SCORE: return p->t_score;
        case IPKGPU_T_COMPACT: return p->t_compact;
        case IPKGPU_T_SCORE_LAUNCHES: return (double)p->score_launches;
        case IPKGPU_T_SCORE_MAIN: return p->t_main;
        case IPKGPU_T_SCORE_REDUCE: return p->t_reduce;
        case IPKGPU_T_XP_COUNT: return p->t_count;
        case IPKGPU_T_XP_WRITE: return p->t_write;
        case IPKGPU_T_KM_WRITE: return p->t_km;
    }
    return 0;
}
void ipkgpu_parts_free(ipkgpu_parts* p)
{
    if (!p) return;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        ctx_release(p->ctx, p->d_counts); ctx_release(p->ctx, p->d_entries); ctx_release(p->ctx, p->pre_keys); ctx_release(p->ctx, p->pre_key_off);
    }
    delete p;
}

int ipkgpu_merge_parts(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t n_owners, uint32_t n_sources,
                       const uint32_t* counts_dev, const void* entries_dev, const uint64_t* source_offsets, ipkgpu_db** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!counts_dev || !source_offsets) return fail(ctx, IPKGPU_ERR_INVALID, "null input pointer");
    if ((sigma != 4 && sigma != 20) || k < 2 || k > ipkgpu_max_k(sigma)) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
    if (n_owners == 0 || owner >= n_owners || n_sources == 0) return fail(ctx, IPKGPU_ERR_INVALID, "bad owner/source counts");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t T = ipow(sigma, (int)k);
    const uint64_t slots = (T + n_owners - 1) / n_owners;
    ipkgpu_db* db = new (std::nothrow) ipkgpu_db();
    if (!db) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    db->ctx = ctx;
    struct Guard { ipkgpu_db* r; ~Guard() { if (r) ipkgpu_db_free(r); } } guard{db};
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t0 = sw.mark();
    std::vector<const uint32_t*> crow(n_sources);
    std::vector<const uint2*> srow(n_sources);
    for (uint32_t s = 0; s < n_sources; ++s) {
        crow[s] = counts_dev + (size_t)s * slots;
        srow[s] = reinterpret_cast<const uint2*>(entries_dev) + source_offsets[s];
    }
    RC_TRY(ensure(ctx, ctx->counts, slots * 4));
    RC_TRY(merge_sources(ctx, sigma, k, owner, n_owners, n_sources, slots, crow, srow, ctx->counts.as<uint32_t>(), &db->d_entries, &db->n_entries, db));
    const int t1 = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    db->t_merge = sw.ms(t0, t1);
    guard.r = nullptr;
    *out = db;
    return IPKGPU_OK;
}

int ipkgpu_merge_parts_ptrs(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t n_owners, uint32_t n_sources,
                            const uint32_t* const* counts_dev, const void* const* entries_dev, ipkgpu_db** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!counts_dev || !entries_dev) return fail(ctx, IPKGPU_ERR_INVALID, "null input pointer");
    if ((sigma != 4 && sigma != 20) || k < 2 || k > ipkgpu_max_k(sigma)) return fail(ctx, IPKGPU_ERR_INVALID, "unsupported sigma/k");
    if (n_owners == 0 || owner >= n_owners || n_sources == 0) return fail(ctx, IPKGPU_ERR_INVALID, "bad owner/source counts");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t T = ipow(sigma, (int)k);
    const uint64_t slots = (T + n_owners - 1) / n_owners;
    ipkgpu_db* db = new (std::nothrow) ipkgpu_db();
    if (!db) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    db->ctx = ctx;
    struct Guard { ipkgpu_db* r; ~Guard() { if (r) ipkgpu_db_free(r); } } guard{db};
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t0 = sw.mark();
    std::vector<const uint32_t*> crow(counts_dev, counts_dev + n_sources);
    std::vector<const uint2*> srow(n_sources);
    for (uint32_t s = 0; s < n_sources; ++s) srow[s] = reinterpret_cast<const uint2*>(entries_dev[s]);
    RC_TRY(ensure(ctx, ctx->counts, slots * 4));
    RC_TRY(merge_sources(ctx, sigma, k, owner, n_owners, n_sources, slots, crow, srow, ctx->counts.as<uint32_t>(), &db->d_entries, &db->n_entries, db));
    const int t1 = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    db->t_merge = sw.ms(t0, t1);
    guard.r = nullptr;
    *out = db;
    return IPKGPU_OK;
}

int ipkgpu_db_from_parts(ipkgpu_ctx* ctx, ipkgpu_parts* parts, uint32_t sigma, uint32_t k, ipkgpu_db** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!parts || parts->n_owners != 1 || !parts->d_counts) return fail(ctx, IPKGPU_ERR_INVALID, "parts must be single-owner and not yet consumed");
    if ((sigma != 4 && sigma != 20) || k < 2 || k > ipkgpu_max_k(sigma) || parts->slots != ipow(sigma, (int)k))
        return fail(ctx, IPKGPU_ERR_INVALID, "sigma/k do not match the parts");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ipkgpu_db* db = new (std::nothrow) ipkgpu_db();
    if (!db) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    db->ctx = ctx;
    struct Guard { ipkgpu_db* r; ~Guard() { if (r) ipkgpu_db_free(r); } } guard{db};
    const uint64_t slots = parts->slots;
    if (parts->pre_keys) {                     // the scoring call built the key list already
        db->d_keys = parts->pre_keys; db->d_key_off = parts->pre_key_off; db->n_keys = parts->pre_n_keys;
        parts->pre_keys = nullptr; parts->pre_key_off = nullptr;
        db->n_entries = parts->owner_off[1];
        db->d_entries = parts->d_entries;      // ownership moves to the database
        parts->d_entries = nullptr;
        db->t_merge = parts->t_keys;
        guard.r = nullptr;
        *out = db;
        return IPKGPU_OK;
    }
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t0 = sw.mark();
    // single source, single owner: the entries are already in database order; only the key list
    // (non-empty slots) and its offsets have to be produced
    RC_TRY(ensure(ctx, ctx->tmp_a, slots * 4));
    RC_TRY(ensure(ctx, ctx->tmp_b, (slots + 1) * 8));
    RC_TRY(ensure(ctx, ctx->counts, slots * 4));
    RC_TRY(ensure(ctx, ctx->offsets, (slots + 1) * 8));
    hipLaunchKernelGGL(flags_from_counts_kernel, dim3((uint32_t)((slots + 255) / 256)), dim3(256), 0, ctx->stream,
                       parts->d_counts, slots, ctx->counts.as<uint32_t>(), ctx->tmp_a.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError());
    RC_TRY(scan_u32(ctx, parts->d_counts, slots, ctx->tmp_b.as<uint64_t>()));
    RC_TRY(scan_u32(ctx, ctx->tmp_a.as<uint32_t>(), slots, ctx->offsets.as<uint64_t>()));
    // (no wait for the number of keys: it cannot exceed the slots nor the entries, the key arrays are allocated for that bound --
    //  at most 12 bytes per slot next to 8 bytes per entry -- and the count itself comes back with the call's last wait)
    const uint64_t keys_bound = std::max<uint64_t>(1, std::min<uint64_t>(slots, parts->owner_off[1]));
    uint64_t* h_keys = reinterpret_cast<uint64_t*>(ctx->h_rb) + RB_OWNER_OFF;
    HIP_TRY(ctx, hipMemcpyAsync(h_keys, ctx->offsets.as<uint64_t>() + slots, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_keys, keys_bound * 4));
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_key_off, (keys_bound + 1) * 8));
    const uint32_t nbk = (uint32_t)((slots + 1 + 255) / 256);
    if (sigma == 4)
        hipLaunchKernelGGL(merge_write_keys_kernel<4>, dim3(nbk), dim3(256), 0, ctx->stream, ctx->counts.as<uint32_t>(),
                           ctx->offsets.as<uint64_t>(), ctx->tmp_b.as<uint64_t>(), slots, 0u, 1u, (int)k, db->d_keys, db->d_key_off);
    else
        hipLaunchKernelGGL(merge_write_keys_kernel<20>, dim3(nbk), dim3(256), 0, ctx->stream, ctx->counts.as<uint32_t>(),
                           ctx->offsets.as<uint64_t>(), ctx->tmp_b.as<uint64_t>(), slots, 0u, 1u, (int)k, db->d_keys, db->d_key_off);
    HIP_TRY(ctx, hipGetLastError());
    const int t1 = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    db->n_keys = *h_keys;
    db->n_entries = parts->owner_off[1];
    db->d_entries = parts->d_entries;          // ownership moves to the database
    parts->d_entries = nullptr;
    db->t_merge = sw.ms(t0, t1);
    guard.r = nullptr;
    *out = db;
    return IPKGPU_OK;
}

uint64_t ipkgpu_db_num_keys(const ipkgpu_db* d) { return d ? d->n_keys : 0; }
uint64_t ipkgpu_db_num_entries(const ipkgpu_db* d) { return d ? d->n_entries : 0; }
const uint32_t* ipkgpu_db_keys_device(const ipkgpu_db* d) { return d ? d->d_keys : nullptr; }
const uint64_t* ipkgpu_db_key_offsets_device(const ipkgpu_db* d) { return d ? d->d_key_off : nullptr; }
const void* ipkgpu_db_entries_device(const ipkgpu_db* d) { return d ? d->d_entries : nullptr; }
double ipkgpu_db_time_ms(const ipkgpu_db* d) { return d ? d->t_merge : 0; }

static bool db_to_host(ipkgpu_db* d)
{
    if (d->h_ok) return true;
    (void)hipSetDevice(d->ctx->device);
    d->h_keys.resize(std::max<uint64_t>(d->n_keys, 1));
    d->h_key_off.resize(d->n_keys + 1);
    d->h_entries.resize(std::max<uint64_t>(d->n_entries, 1) * 2);
    if (d->n_keys && hipMemcpy(d->h_keys.data(), d->d_keys, d->n_keys * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(d->h_key_off.data(), d->d_key_off, (d->n_keys + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (d->n_entries && hipMemcpy(d->h_entries.data(), d->d_entries, d->n_entries * 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
    d->h_ok = true;
    return true;
}
const uint32_t* ipkgpu_db_keys(ipkgpu_db* d) { return d && db_to_host(d) ? d->h_keys.data() : nullptr; }
const uint64_t* ipkgpu_db_key_offsets(ipkgpu_db* d) { return d && db_to_host(d) ? d->h_key_off.data() : nullptr; }
const uint32_t* ipkgpu_db_entries(ipkgpu_db* d) { return d && db_to_host(d) ? d->h_entries.data() : nullptr; }

float ipkgpu_score_threshold(float omega, uint32_t sigma, uint32_t k) { return powf(omega / (float)sigma, (float)k); }

int ipkgpu_db_filter_mif0(ipkgpu_ctx* ctx, ipkgpu_db* db, uint64_t total_num_groups, float threshold)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!db || db->ctx != ctx) return fail(ctx, IPKGPU_ERR_INVALID, "database does not belong to this context");
    if (total_num_groups == 0 || !(threshold > 0.0f)) return fail(ctx, IPKGPU_ERR_INVALID, "need total_num_groups > 0 and threshold > 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = db->n_keys;
    ctx_release(ctx, db->d_fv64); ctx_release(ctx, db->d_fv32); ctx_release(ctx, db->d_order);
    db->d_fv64 = nullptr; db->d_fv32 = nullptr; db->d_order = nullptr; db->h_filter_ok = false;
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_fv64, std::max<uint64_t>(n, 1) * 8));
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_fv32, std::max<uint64_t>(n, 1) * 4));
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&db->d_order, std::max<uint64_t>(n, 1) * 4));
    if (n == 0) return IPKGPU_OK;
    Stopwatch sw(ctx->stream, &ctx->events);
    const int t0 = sw.mark();
    for (uint64_t first = 0; first < n; first += WAVE_PER_ITEM_SPAN)
        hipLaunchKernelGGL(mif0_kernel, dim3((uint32_t)((std::min<uint64_t>(WAVE_PER_ITEM_SPAN, n - first) + 3) / 4)), dim3(256), 0, ctx->stream,
                           db->d_key_off, db->d_entries, n, (double)total_num_groups, (double)threshold, db->d_fv64, db->d_fv32, first);
    HIP_TRY(ctx, hipGetLastError());
    // order: ascending filter value (std::sort of kmer_order, db_builder.cpp:284), ties by ascending key
    RC_TRY(ensure(ctx, ctx->tmp_a, n * 8));
    RC_TRY(ensure(ctx, ctx->tmp_b, n * 8));
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(filter_sortkey_kernel, dim3(nb), dim3(256), 0, ctx->stream, db->d_fv32, n, ctx->tmp_a.as<unsigned long long>());
    HIP_TRY(ctx, hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, rocprim::radix_sort_keys(nullptr, tmp_bytes, ctx->tmp_a.as<unsigned long long>(), ctx->tmp_b.as<unsigned long long>(),
                                          (size_t)n, 0, 64, ctx->stream));
    RC_TRY(ensure(ctx, ctx->tmp_c, tmp_bytes));
    HIP_TRY(ctx, rocprim::radix_sort_keys(ctx->tmp_c.p, tmp_bytes, ctx->tmp_a.as<unsigned long long>(), ctx->tmp_b.as<unsigned long long>(),
                                          (size_t)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(filter_order_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->tmp_b.as<unsigned long long>(), n, db->d_order);
    HIP_TRY(ctx, hipGetLastError());
    const int t1 = sw.mark();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    db->t_filter = sw.ms(t0, t1);
    return IPKGPU_OK;
}

static bool db_filter_to_host(ipkgpu_db* d)
{
    if (d->h_filter_ok) return true;
    if (!d->d_fv64) return false;
    (void)hipSetDevice(d->ctx->device);
    const uint64_t n = d->n_keys;
    d->h_fv64.resize(std::max<uint64_t>(n, 1)); d->h_fv32.resize(std::max<uint64_t>(n, 1)); d->h_order.resize(std::max<uint64_t>(n, 1));
    if (n) {
        if (hipMemcpy(d->h_fv64.data(), d->d_fv64, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(d->h_fv32.data(), d->d_fv32, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(d->h_order.data(), d->d_order, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    }
    d->h_filter_ok = true;
    return true;
}
const double* ipkgpu_db_filter_values_f64(ipkgpu_db* d) { return d && db_filter_to_host(d) ? d->h_fv64.data() : nullptr; }
const float* ipkgpu_db_filter_values(ipkgpu_db* d) { return d && db_filter_to_host(d) ? d->h_fv32.data() : nullptr; }
const uint32_t* ipkgpu_db_filter_order(ipkgpu_db* d) { return d && db_filter_to_host(d) ? d->h_order.data() : nullptr; }
const float* ipkgpu_db_filter_values_device(const ipkgpu_db* d) { return d ? d->d_fv32 : nullptr; }
const uint32_t* ipkgpu_db_filter_order_device(const ipkgpu_db* d) { return d ? d->d_order : nullptr; }
double ipkgpu_db_filter_time_ms(const ipkgpu_db* d) { return d ? d->t_filter : 0; }

void ipkgpu_db_free(ipkgpu_db* d)
{
    if (!d) return;
    if (d->ctx) {
        (void)hipSetDevice(d->ctx->device);
        ctx_release(d->ctx, d->d_keys); ctx_release(d->ctx, d->d_key_off); ctx_release(d->ctx, d->d_entries);
        ctx_release(d->ctx, d->d_fv64); ctx_release(d->ctx, d->d_fv32); ctx_release(d->ctx, d->d_order);
    }
    delete d;
}

}  // extern "C"

// ---- "next" row n2: the database file -------------------------------------------------------------------
namespace {
thread_local std::string g_write_err;

struct FileOut {
    FILE* f = nullptr;
    ~FileOut() { if (f) fclose(f); }
    bool put(const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
};
}  // namespace

extern "C" {

const char* ipkgpu_db_write_last_error(void) { return g_write_err.c_str(); }
const char* ipkgpu_last_main_kernel(const ipkgpu_ctx* ctx) { return ctx ? ctx->main_kernel : ""; }
int ipkgpu_last_tables_compressed(const ipkgpu_ctx* ctx) { return ctx && ctx->table_compressed ? 1 : 0; }

// Diagnostics: calls of the exec-writing asm helpers entered with a partial exec mask since the library was loaded
// (IPK_EXEC_ASSERT builds only; -1 in the shipped build, which compiles the check away).
int64_t ipkgpu_debug_exec_violations(ipkgpu_ctx* ctx)
{
#ifdef IPK_EXEC_ASSERT
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -2;
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(ipkgpu::g_exec_violations), sizeof v) != hipSuccess) return -2;
    return (int64_t)v;
#else
    (void)ctx;
    return -1;
#endif
}
double ipkgpu_db_write_time_s(const ipkgpu_ctx* ctx, int which)
{
    if (!ctx) return 0;
    return which == 1 ? ctx->t_write_device : which == 2 ? ctx->t_write_file : ctx->t_write_total;
}

int ipkgpu_db_write_host(const ipkgpu_db_header* h, uint64_t n_keys, const uint32_t* keys, const uint64_t* key_off, const uint32_t* entries,
                         const float* fv, const uint32_t* order, const char* path, uint64_t* bytes_written)
{
    if (!h || !path || (n_keys && (!keys || !key_off || !fv))) { g_write_err = "null argument"; return IPKGPU_ERR_INVALID; }
    FileOut out;
    out.f = fopen(path, "wb");
    if (!out.f) { g_write_err = std::string("cannot create ") + path; return IPKGPU_ERR_INVALID; }
    const uint64_t n_entries = n_keys ? key_off[n_keys] : 0;
    const std::vector<uint8_t> head = ipkfmt::file_head(h->sequence_type, h->tree_index_size, h->tree_num_nodes, h->tree_subtree_length, h->newick,
                                                        h->kmer_size, h->omega, n_keys, n_entries);
    uint64_t total = head.size();
    if (!out.put(head.data(), head.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; }
    std::vector<uint8_t> buf;
    buf.reserve((size_t)16 << 20);
    for (uint64_t i = 0; i < n_keys; ++i) {
        const uint64_t k = order ? order[i] : i;
        const uint64_t a = key_off[k], n = key_off[k + 1] - a;
        uint32_t w[4];
        uint32_t fb; memcpy(&fb, &fv[k], 4);
        ipkfmt::record_head(keys[k], fb, n, w);
        ipkfmt::put(buf, w, sizeof w);
        ipkfmt::put(buf, entries + 2 * a, (size_t)(n * ipkfmt::ENTRY_BYTES));
        if (buf.size() >= ((size_t)15 << 20)) { if (!out.put(buf.data(), buf.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; } total += buf.size(); buf.clear(); }
    }
    if (!out.put(buf.data(), buf.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; }
    total += buf.size();
    if (fclose(out.f) != 0) { out.f = nullptr; g_write_err = "close failed"; return IPKGPU_ERR_INVALID; }
    out.f = nullptr;
    if (bytes_written) *bytes_written = total;
    return IPKGPU_OK;
}

// The positioned database (ipk-aa-pos: every entry carries the window position of its kept score, db_builder.cpp:655-662,687-689;
// branch_group.cpp:73-86): header with the positions flag set, entries of (branch, score, position).
int ipkgpu_db_write_host_positions(const ipkgpu_db_header* h, uint64_t n_keys, const uint32_t* keys, const uint64_t* key_off,
                                   const uint32_t* entries, const uint32_t* positions, const float* fv, const uint32_t* order,
                                   const char* path, uint64_t* bytes_written)
{
    if (!h || !path || (n_keys && (!keys || !key_off || !fv || !entries || !positions))) { g_write_err = "null argument"; return IPKGPU_ERR_INVALID; }
    if (ipkfmt::protocol_version() == 0) { g_write_err = "a positioned database needs the positions flag: IPKGPU_IPK_PROTOCOL_VERSION must not be 0"; return IPKGPU_ERR_INVALID; }
    const uint64_t n_entries = n_keys ? key_off[n_keys] : 0;
    for (uint64_t i = 0; i < n_entries; ++i)
        if (positions[i] > 0xFFFFu) { g_write_err = "a window position beyond 65535 does not fit the entry's position field"; return IPKGPU_ERR_INVALID; }
    FileOut out;
    out.f = fopen(path, "wb");
    if (!out.f) { g_write_err = std::string("cannot create ") + path; return IPKGPU_ERR_INVALID; }
    const std::vector<uint8_t> head = ipkfmt::file_head(h->sequence_type, h->tree_index_size, h->tree_num_nodes, h->tree_subtree_length, h->newick,
                                                        h->kmer_size, h->omega, n_keys, n_entries, true);
    uint64_t total = head.size();
    if (!out.put(head.data(), head.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; }
    std::vector<uint8_t> buf;
    buf.reserve((size_t)16 << 20);
    for (uint64_t i = 0; i < n_keys; ++i) {
        const uint64_t k = order ? order[i] : i;
        const uint64_t a = key_off[k], n = key_off[k + 1] - a;
        uint32_t w[4];
        uint32_t fb; memcpy(&fb, &fv[k], 4);
        ipkfmt::record_head(keys[k], fb, n, w);
        ipkfmt::put(buf, w, sizeof w);
        for (uint64_t j = a; j < a + n; ++j) {
            ipkfmt::put(buf, entries + 2 * j, 8);
            ipkfmt::put_v<uint16_t>(buf, (uint16_t)positions[j]);
        }
        if (buf.size() >= ((size_t)15 << 20)) { if (!out.put(buf.data(), buf.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; } total += buf.size(); buf.clear(); }
    }
    if (!out.put(buf.data(), buf.size())) { g_write_err = "write failed"; return IPKGPU_ERR_INVALID; }
    total += buf.size();
    if (fclose(out.f) != 0) { out.f = nullptr; g_write_err = "close failed"; return IPKGPU_ERR_INVALID; }
    out.f = nullptr;
    if (bytes_written) *bytes_written = total;
    return IPKGPU_OK;
}

int ipkgpu_db_write(ipkgpu_ctx* ctx, ipkgpu_db* db, const ipkgpu_db_header* h, const char* path, uint64_t* bytes_written)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!db || db->ctx != ctx || !h || !path) return fail(ctx, IPKGPU_ERR_INVALID, "bad argument");
    if (db->n_keys && (!db->d_fv32 || !db->d_order)) return fail(ctx, IPKGPU_ERR_INVALID, "filter values missing: call ipkgpu_db_filter_mif0 first");
    if (db->n_keys >= 0xFFFFFFFFull) return fail(ctx, IPKGPU_ERR_INVALID, "too many k-mers");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    double t_dev = 0, t_file = 0;
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    FileOut out;
    out.f = fopen(path, "wb");
    if (!out.f) return fail(ctx, IPKGPU_ERR_INVALID, "cannot create %s", path);
    setvbuf(out.f, nullptr, _IONBF, 0);                              // the pieces are large: no second copy through stdio
    const uint64_t n = db->n_keys;
    const std::vector<uint8_t> head = ipkfmt::file_head(h->sequence_type, h->tree_index_size, h->tree_num_nodes, h->tree_subtree_length, h->newick,
                                                        h->kmer_size, h->omega, n, db->n_entries);
    uint64_t total = head.size();
    { const auto t0 = std::chrono::steady_clock::now(); if (!out.put(head.data(), head.size())) return fail(ctx, IPKGPU_ERR_INVALID, "write failed"); t_file += since(t0); }
    if (n) {
        // record offsets in filter order
        RC_TRY(ensure(ctx, ctx->tmp_a, n * 4));
        RC_TRY(ensure(ctx, ctx->tmp_b, (n + 1) * 8));
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t* d_big = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ctx->small) + 60);
        HIP_TRY(ctx, hipMemsetAsync(d_big, 0, 4, ctx->stream));
        hipLaunchKernelGGL(db_record_sizes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, db->d_order, db->d_key_off, n,
                           ctx->tmp_a.as<uint32_t>(), d_big);
        HIP_TRY(ctx, hipGetLastError());
        RC_TRY(scan_u32(ctx, ctx->tmp_a.as<uint32_t>(), n, ctx->tmp_b.as<uint64_t>()));
        std::vector<uint64_t> rec_off(n + 1);
        uint32_t h_big = 0;
        HIP_TRY(ctx, hipMemcpyAsync(rec_off.data(), ctx->tmp_b.p, (n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&h_big, d_big, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h_big) return fail(ctx, IPKGPU_ERR_INVALID, "a k-mer with 2^28 entries or more: record sizes are 32-bit");
        t_dev += since(t0);
        // pieces of whole records, two staging buffers: piece j+1 is packed and copied while piece j is written
        uint64_t max_rec = 0;
        for (uint64_t i = 0; i < n; ++i) max_rec = std::max(max_rec, rec_off[i + 1] - rec_off[i]);
        const uint64_t PIECE = std::max<uint64_t>((uint64_t)256 << 20, max_rec);
        void* d_stage[2] = {nullptr, nullptr};
        void* h_stage[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        struct Cleanup { ipkgpu_ctx* c; void** d; void** h; hipEvent_t* e;
            ~Cleanup() { for (int i = 0; i < 2; ++i) { ctx_release(c, d[i]); if (h[i]) (void)hipHostFree(h[i]); if (e[i]) (void)hipEventDestroy(e[i]); } } } cleanup{ctx, d_stage, h_stage, ev};
        const uint64_t body = rec_off[n];
        const uint64_t stage_bytes = std::min<uint64_t>(PIECE, body);
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(ctx, ctx_alloc(ctx, &d_stage[i], std::max<uint64_t>(stage_bytes, 16)));
            HIP_TRY(ctx, hipHostMalloc(&h_stage[i], std::max<uint64_t>(stage_bytes, 16), hipHostMallocDefault));
            HIP_TRY(ctx, hipEventCreate(&ev[i]));
        }
        struct Piece { uint64_t lo, hi; };
        std::vector<Piece> pieces;
        for (uint64_t lo = 0; lo < n;) {
            // largest hi with rec_off[hi] - rec_off[lo] <= PIECE
            const uint64_t hi = (uint64_t)(std::upper_bound(rec_off.begin() + lo, rec_off.end(), rec_off[lo] + PIECE) - rec_off.begin()) - 1;
            pieces.push_back({lo, std::max(hi, lo + 1)});
            lo = pieces.back().hi;
        }
        auto launch = [&](size_t j) -> int {
            const Piece pc = pieces[j];
            const int b = (int)(j & 1);
            hipLaunchKernelGGL(db_pack_kernel, dim3((uint32_t)((pc.hi - pc.lo + 3) / 4)), dim3(256), 0, ctx->stream, db->d_order, db->d_keys, db->d_key_off,
                               db->d_entries, db->d_fv32, ctx->tmp_b.as<uint64_t>(), pc.lo, pc.hi, (unsigned char*)d_stage[b]);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(h_stage[b], d_stage[b], rec_off[pc.hi] - rec_off[pc.lo], hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ev[b], ctx->stream));
            return IPKGPU_OK;
        };
        if (!pieces.empty()) RC_TRY(launch(0));
        for (size_t j = 0; j < pieces.size(); ++j) {
            if (j + 1 < pieces.size()) RC_TRY(launch(j + 1));
            const auto tw = std::chrono::steady_clock::now();
            HIP_TRY(ctx, hipEventSynchronize(ev[j & 1]));
            t_dev += since(tw);
            const auto tf = std::chrono::steady_clock::now();
            const uint64_t nb = rec_off[pieces[j].hi] - rec_off[pieces[j].lo];
            if (!out.put(h_stage[j & 1], nb)) return fail(ctx, IPKGPU_ERR_INVALID, "write failed");
            t_file += since(tf);
            total += nb;
        }
    }
    { const auto t0 = std::chrono::steady_clock::now(); const int rc = fclose(out.f); out.f = nullptr; if (rc != 0) return fail(ctx, IPKGPU_ERR_INVALID, "close failed"); t_file += since(t0); }
    ctx->t_write_total = since(t_begin); ctx->t_write_device = t_dev; ctx->t_write_file = t_file;
    if (bytes_written) *bytes_written = total;
    return IPKGPU_OK;
}

}  // extern "C"

#include "comm_rccl.hpp"
