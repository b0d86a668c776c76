// comm.cpp -- the one collective of the path: an all_gather of per-GPU counters over RCCL (xGMI inside a node).
//
// Reads shard by record and nothing is exchanged on the data path; what a job reports at its
// end -- reads, bases, tuples, flagged reads per GPU -- is gathered here so that the host language above the C ABI (Go in
// the reference's world, Python in this repository's tests and bench) needs no collective library of its own.
// librccl is opened on first use (dlopen): a single-GPU caller never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "host_types.hpp"

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *why = nullptr;  // dlerror() clears its state when read: call it once, right after the failing dlopen
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
            why = dlerror();
        }
        if (!r.h) {
            r.err = std::string("librccl not found: ") + (why ? why : "");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.h, n);
            if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int fail_nccl(bsk_ctx *ctx, ncclResult_t e, const char *what) {
    Rccl *r = rccl();
    if (ctx) ctx->err = std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
    return BSK_ERR_DEVICE;
}

int need_rccl(bsk_ctx *ctx) {
    Rccl *r = rccl();
    if (!r->h || !r->err.empty()) {
        if (ctx) ctx->err = r->err;
        return BSK_ERR_DEVICE;
    }
    return BSK_OK;
}

int ensure_buf(bsk_ctx *ctx, int world) {
    if (ctx->d_comm) return BSK_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(&ctx->d_comm, (size_t)(world + 1) * BSK_MAX_COUNTERS * sizeof(u64)));
    return BSK_OK;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == BSK_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int bsk_comm_unique_id(uint8_t *id) {
    if (!id) return BSK_ERR_ARG;
    if (need_rccl(nullptr) != BSK_OK) return BSK_ERR_DEVICE;
    ncclUniqueId u;
    if (rccl()->GetUniqueId(&u) != ncclSuccess) return BSK_ERR_DEVICE;
    memcpy(id, &u, sizeof u);
    return BSK_OK;
}

extern "C" int bsk_comm_init_rank(bsk_ctx *ctx, const uint8_t *id, int rank, int world) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return fail_arg(ctx, "bsk_comm_init_rank: bad argument");
    if (ctx->comm) return fail_arg(ctx, "bsk_comm_init_rank: the context already belongs to a communicator");
    int rc = need_rccl(ctx);
    if (rc != BSK_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    const ncclResult_t e = rccl()->CommInitRank(&c, world, u, rank);
    if (e != ncclSuccess) return fail_nccl(ctx, e, "ncclCommInitRank");
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return ensure_buf(ctx, world);
}

extern "C" int bsk_comm_init_all(bsk_ctx *const *ctxs, int n) {
    if (!ctxs || n < 1) return BSK_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->comm) return fail_arg(ctxs[i], "bsk_comm_init_all: null context or context already in a communicator");
    int rc = need_rccl(ctxs[0]);
    if (rc != BSK_OK) return rc;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
    const ncclResult_t e = rccl()->CommInitAll(comms.data(), n, devs.data());
    if (e != ncclSuccess) return fail_nccl(ctxs[0], e, "ncclCommInitAll");
    for (int i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = n;
        rc = ensure_buf(ctxs[i], n);
        if (rc != BSK_OK) return rc;
    }
    return BSK_OK;
}

// one rank's part: counters to the device, the all_gather on the context's stream
static int gather_enqueue(bsk_ctx *ctx, const uint64_t *mine, int nc) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    u64 *send = ctx->d_comm, *recv = ctx->d_comm + BSK_MAX_COUNTERS;
    HIPCHK(ctx, hipMemcpyAsync(send, mine, (size_t)nc * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t e = rccl()->AllGather(send, recv, (size_t)nc, ncclUint64, (ncclComm_t)ctx->comm, ctx->stream);
    if (e != ncclSuccess) return fail_nccl(ctx, e, "ncclAllGather");
    return BSK_OK;
}
static int gather_finish(bsk_ctx *ctx, int nc, uint64_t *all) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(all, ctx->d_comm + BSK_MAX_COUNTERS, (size_t)nc * ctx->comm_world * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BSK_OK;
}

extern "C" int bsk_gather_counts(bsk_ctx *ctx, const uint64_t *mine, int n_counters, uint64_t *all) {
    if (!ctx || !mine || !all || n_counters < 1 || n_counters > BSK_MAX_COUNTERS) return fail_arg(ctx, "bsk_gather_counts: bad argument");
    if (!ctx->comm) return fail_arg(ctx, "bsk_gather_counts: no communicator (bsk_comm_init_rank / bsk_comm_init_all first)");
    int rc = gather_enqueue(ctx, mine, n_counters);
    if (rc != BSK_OK) return rc;
    return gather_finish(ctx, n_counters, all);
}

extern "C" int bsk_gather_counts_all(bsk_ctx *const *ctxs, int n, const uint64_t *mine, int n_counters, uint64_t *all) {
    if (!ctxs || n < 1 || !mine || !all || n_counters < 1 || n_counters > BSK_MAX_COUNTERS) return BSK_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->comm_world != n || ctxs[i]->comm_rank != i)
            return fail_arg(ctxs[i], "bsk_gather_counts_all: the contexts are not the ranks 0..n-1 of one bsk_comm_init_all communicator");
    // one thread drives every rank: the collective calls must sit in one group
    ncclResult_t e = rccl()->GroupStart();
    if (e != ncclSuccess) return fail_nccl(ctxs[0], e, "ncclGroupStart");
    int rc = BSK_OK;
    for (int i = 0; i < n && rc == BSK_OK; ++i) rc = gather_enqueue(ctxs[i], mine + (size_t)i * n_counters, n_counters);
    e = rccl()->GroupEnd();
    if (rc != BSK_OK) return rc;
    if (e != ncclSuccess) return fail_nccl(ctxs[0], e, "ncclGroupEnd");
    std::vector<uint64_t> other((size_t)n * n_counters);
    for (int i = 0; i < n; ++i) {
        rc = gather_finish(ctxs[i], n_counters, i == 0 ? all : other.data());
        if (rc != BSK_OK) return rc;
        if (i && memcmp(other.data(), all, other.size() * sizeof(uint64_t)) != 0) {
            ctxs[i]->err = "bsk_gather_counts_all: ranks received different data";
            return BSK_ERR_DEVICE;
        }
    }
    return BSK_OK;
}

extern "C" void bsk_comm_destroy(bsk_ctx *ctx) {
    if (!ctx) return;
    if (ctx->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    if (ctx->d_comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipFree(ctx->d_comm);
        ctx->d_comm = nullptr;
    }
}
