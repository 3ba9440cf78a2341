// comm_rccl.hpp -- the k-mer-keyed exchange of database parts over RCCL, inside the library (included by ipkgpu.hip).
//
// CPU analogue in the reference: the on-disk path's k-mer-keyed partition and merge -- kmer_batch
// (ipk/src/branch_group.cpp:104-107), merge_batch (:45-70), merge_stage2 (ipk/src/db_builder.cpp:392-458).  Here rank r scores
// a contiguous range of branch groups, ipkgpu_score_groups_keymajor_device(..., n_owners = world) leaves its entries split by
// owner, and block o travels to rank o: grouped ncclSend / ncclRecv, one pair per peer (direct xGMI links, no ring).
//
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy the process already holds, e.g. PyTorch's, is reused), so a
// single-GPU build never touches it and a missing library is an error code, not a load failure.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional: tears a communicator down without a collective
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("RCCL symbol missing: ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
RcclApi g_rccl;

}  // namespace

struct ipkgpu_comm {
    ncclComm_t comm = nullptr;            // null while only prepared (ipkgpu_comm_prepare) or after a failed exchange
    bool dead = false;                    // a collective of this communicator failed on this rank: aborted, unusable
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;         // the exchange runs beside the scoring stream
    uint64_t* d_sizes = nullptr;          // [2][world]: entries sent to / received from every peer
};

// One piece's exchange in flight.
struct ipkgpu_xfer {
    ipkgpu_ctx* ctx = nullptr;
    ipkgpu_parts* parts = nullptr;        // the sender's blocks stay alive until the merge
    uint64_t slots = 0;
    uint32_t* d_rcounts = nullptr;        // [world][slots]: row s = source rank s's counts of my keys
    uint2* d_rentries = nullptr;          // sources' blocks back to back
    std::vector<uint64_t> roff;           // [world + 1] entry offsets of the sources' blocks
    hipEvent_t done = nullptr;
    double t_exposed_ms = 0;
};

#define NCCL_TRY(ctx, expr)                                                                                 \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(ctx, IPKGPU_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

// the entry ranges of the owners' blocks, cut where the transfer has to be cut: every rank's block o of `owner_off`
static inline void exchange_split(const uint64_t* owner_off, uint32_t world, uint64_t* send_counts)
{
    for (uint32_t o = 0; o < world; ++o) send_counts[o] = owner_off[o + 1] - owner_off[o];
}

static void ipkgpu_comm_release(ipkgpu_ctx* ctx)
{
    ipkgpu_comm* c = ctx->comm;
    if (!c) return;
    if (c->stream && !c->dead) { (void)hipStreamSynchronize(c->stream); }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    delete c;
    ctx->comm = nullptr;
}

extern "C" {

int ipkgpu_comm_unique_id(uint8_t* id128)
{
    if (!id128) return IPKGPU_ERR_INVALID;
    if (!g_rccl.load()) return fail(nullptr, IPKGPU_ERR_NODEVICE, "%s", g_rccl.err.c_str());
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, IPKGPU_ERR_HIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    static_assert(sizeof id.internal == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, id.internal, 128);
    return IPKGPU_OK;
}

// Everything of the communicator set-up that can fail on ONE rank alone -- loading RCCL, the exchange stream, the size
// buffers -- happens here, BEFORE the collective ncclCommInitRank: the ranks agree that every one of them got this far (the
// caller all-reduces the return codes) and only then enter ipkgpu_comm_init, so no rank is left waiting inside the collective
// for a peer that has already given up.
int ipkgpu_comm_available(void)
{
    if (!g_rccl.load()) return fail(nullptr, IPKGPU_ERR_NODEVICE, "%s", g_rccl.err.c_str());
    return IPKGPU_OK;
}

int ipkgpu_comm_prepare(ipkgpu_ctx* ctx, int world)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (world < 1) return fail(ctx, IPKGPU_ERR_INVALID, "bad communicator arguments");
    if (ctx->comm) return ctx->comm->comm ? fail(ctx, IPKGPU_ERR_INVALID, "communicator already initialised") : IPKGPU_OK;
    if (!g_rccl.load()) return fail(ctx, IPKGPU_ERR_NODEVICE, "%s", g_rccl.err.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ipkgpu_comm* c = new (std::nothrow) ipkgpu_comm();
    if (!c) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    c->world = world;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_sizes, (size_t)2 * world * 8);
    if (e != hipSuccess) { if (c->stream) (void)hipStreamDestroy(c->stream); delete c; HIP_TRY(ctx, e); }
    ctx->comm = c;
    return IPKGPU_OK;
}

int ipkgpu_comm_init(ipkgpu_ctx* ctx, const uint8_t* id128, int rank, int world)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, IPKGPU_ERR_INVALID, "bad communicator arguments");
    if (ctx->comm && ctx->comm->comm) return fail(ctx, IPKGPU_ERR_INVALID, "communicator already initialised");
    if (ctx->comm && ctx->comm->world != world) return fail(ctx, IPKGPU_ERR_INVALID, "prepared for another world size");
    const int rp = ipkgpu_comm_prepare(ctx, world);
    if (rp != IPKGPU_OK) return rp;
    ipkgpu_comm* c = ctx->comm;
    c->rank = rank;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        ipkgpu_comm_release(ctx);
        return fail(ctx, IPKGPU_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    return IPKGPU_OK;
}

int ipkgpu_comm_rank(const ipkgpu_ctx* ctx) { return ctx && ctx->comm && ctx->comm->comm ? ctx->comm->rank : 0; }
int ipkgpu_comm_world(const ipkgpu_ctx* ctx) { return ctx && ctx->comm && ctx->comm->comm ? ctx->comm->world : 1; }

void ipkgpu_xfer_free(ipkgpu_xfer* x)
{
    if (!x) return;
    if (x->ctx) {
        (void)hipSetDevice(x->ctx->device);
        if (x->done) { (void)hipEventSynchronize(x->done); (void)hipEventDestroy(x->done); }
        ctx_release(x->ctx, x->d_rcounts); ctx_release(x->ctx, x->d_rentries);
    }
    delete x;
}

// Starts the exchange of one piece: block o of `parts` (counts row o, entries [owner_off[o], owner_off[o+1])) goes to rank o.
// Returns after the transfers have been ENQUEUED on the communicator's stream; `parts` must stay alive until
// ipkgpu_exchange_merge.  The entry counts travel first (one u64 per peer) because the receive sizes must be known on the host.
static int exchange_begin_impl(ipkgpu_ctx* ctx, ipkgpu_comm* c, ipkgpu_parts* parts, ipkgpu_xfer** out);

int ipkgpu_exchange_begin(ipkgpu_ctx* ctx, ipkgpu_parts* parts, ipkgpu_xfer** out)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    ipkgpu_comm* c = ctx->comm;
    if (!c || !c->comm) return fail(ctx, IPKGPU_ERR_INVALID, c && c->dead ? "the communicator was aborted after a failed exchange" : "no communicator: call ipkgpu_comm_init first");
    if (!parts || parts->ctx != ctx || parts->n_owners != (uint32_t)c->world) return fail(ctx, IPKGPU_ERR_INVALID, "parts must be split for n_owners = world size");
    const int rc = exchange_begin_impl(ctx, c, parts, out);
    if (rc != IPKGPU_OK) {
        // a rank that leaves the exchange half way must not leave its peers waiting in the grouped Send/Recv: abort the
        // communicator (their calls then fail instead of hanging); the caller exits non-zero and the launcher ends the job
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm); else (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr; c->dead = true;
    }
    return rc;
}

static int exchange_begin_impl(ipkgpu_ctx* ctx, ipkgpu_comm* c, ipkgpu_parts* parts, ipkgpu_xfer** out)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t P = (uint32_t)c->world;
    ipkgpu_xfer* x = new (std::nothrow) ipkgpu_xfer();
    if (!x) return fail(ctx, IPKGPU_ERR_NOMEM, "out of host memory");
    x->ctx = ctx; x->parts = parts; x->slots = parts->slots;
    struct Guard { ipkgpu_xfer* x; ~Guard() { if (x) ipkgpu_xfer_free(x); } } guard{x};
    // 1. sizes: send_counts[o] to rank o, one u64 each way
    std::vector<uint64_t> sizes(2 * (size_t)P);
    exchange_split(parts->owner_off.data(), P, sizes.data());
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(ctx, hipMemcpyAsync(c->d_sizes, sizes.data(), (size_t)P * 8, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(ctx, g_rccl.GroupStart());
    for (uint32_t o = 0; o < P; ++o) {
        NCCL_TRY(ctx, g_rccl.Send(c->d_sizes + o, 1, ncclUint64, (int)o, c->comm, c->stream));
        NCCL_TRY(ctx, g_rccl.Recv(c->d_sizes + P + o, 1, ncclUint64, (int)o, c->comm, c->stream));
    }
    NCCL_TRY(ctx, g_rccl.GroupEnd());
    HIP_TRY(ctx, hipMemcpyAsync(sizes.data() + P, c->d_sizes + P, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(ctx, hipStreamSynchronize(c->stream));          // also waits for the previous piece's transfer: what scoring did not hide
    x->t_exposed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    x->roff.assign((size_t)P + 1, 0);
    for (uint32_t s = 0; s < P; ++s) x->roff[s + 1] = x->roff[s] + sizes[P + s];
    // 2. payload: counts rows (fixed size) and entry blocks
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&x->d_rcounts, std::max<uint64_t>((uint64_t)P * x->slots, 1) * 4));
    HIP_TRY(ctx, ctx_alloc(ctx, (void**)&x->d_rentries, std::max<uint64_t>(x->roff[P], 1) * 8));
    HIP_TRY(ctx, hipEventCreateWithFlags(&x->done, hipEventDisableTiming));
    NCCL_TRY(ctx, g_rccl.GroupStart());
    for (uint32_t o = 0; o < P; ++o) {
        NCCL_TRY(ctx, g_rccl.Send(parts->d_counts + (size_t)o * x->slots, x->slots, ncclUint32, (int)o, c->comm, c->stream));
        NCCL_TRY(ctx, g_rccl.Recv(x->d_rcounts + (size_t)o * x->slots, x->slots, ncclUint32, (int)o, c->comm, c->stream));
        if (sizes[o]) NCCL_TRY(ctx, g_rccl.Send(parts->d_entries + parts->owner_off[o], sizes[o], ncclUint64, (int)o, c->comm, c->stream));
        if (sizes[P + o]) NCCL_TRY(ctx, g_rccl.Recv(x->d_rentries + x->roff[o], sizes[P + o], ncclUint64, (int)o, c->comm, c->stream));
    }
    NCCL_TRY(ctx, g_rccl.GroupEnd());
    HIP_TRY(ctx, hipEventRecord(x->done, c->stream));
    guard.x = nullptr;
    *out = x;
    return IPKGPU_OK;
}

double ipkgpu_xfer_exposed_ms(const ipkgpu_xfer* x) { return x ? x->t_exposed_ms : 0; }

int ipkgpu_merge_parts_ptrs(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t n_owners, uint32_t n_sources,
                            const uint32_t* const* counts_dev, const void* const* entries_dev, ipkgpu_db** out);

// Waits for the pieces' transfers and merges them into this rank's database shard: sources in the order (rank 0 piece 0,
// rank 0 piece 1, ..., rank 1 piece 0, ...) = global group order, the order the reference appends entries
// (db_builder.cpp:606-618,685-694).  *exposed_ms (optional): time this call waited for transfers.
int ipkgpu_exchange_merge(ipkgpu_ctx* ctx, ipkgpu_xfer* const* xfers, uint32_t n_pieces, uint32_t sigma, uint32_t k, ipkgpu_db** out,
                          double* exposed_ms)
{
    if (!ctx) return IPKGPU_ERR_INVALID;
    if (!out) return fail(ctx, IPKGPU_ERR_INVALID, "null out pointer");
    *out = nullptr;
    ipkgpu_comm* c = ctx->comm;
    if (!c || !c->comm || !xfers || n_pieces == 0) return fail(ctx, IPKGPU_ERR_INVALID, "bad exchange arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t j = 0; j < n_pieces; ++j) {
        if (!xfers[j] || xfers[j]->ctx != ctx) return fail(ctx, IPKGPU_ERR_INVALID, "bad transfer handle");
        HIP_TRY(ctx, hipEventSynchronize(xfers[j]->done));
    }
    if (exposed_ms) {
        *exposed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t j = 0; j < n_pieces; ++j) *exposed_ms += xfers[j]->t_exposed_ms;
    }
    const uint32_t P = (uint32_t)c->world;
    std::vector<const uint32_t*> cp; std::vector<const void*> ep;
    for (uint32_t s = 0; s < P; ++s)
        for (uint32_t j = 0; j < n_pieces; ++j) {
            cp.push_back(xfers[j]->d_rcounts + (size_t)s * xfers[j]->slots);
            ep.push_back(xfers[j]->d_rentries + xfers[j]->roff[s]);
        }
    return ipkgpu_merge_parts_ptrs(ctx, sigma, k, (uint32_t)c->rank, P, (uint32_t)cp.size(), cp.data(), ep.data(), out);
}

}  // extern "C"
