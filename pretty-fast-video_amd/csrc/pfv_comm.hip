// pfv_comm.hip -- the control-plane collectives of the multi-GPU path, straight on RCCL (included by pfv_capi.hip).
//
// The hot path shards by stream (src/enc.rs:12-26: an Encoder shares nothing with another) and by GOP (an i-frame never reads
// prev_frame, src/enc.rs:84-97): no pixel or coefficient ever crosses GPUs.  What does cross is a few hundred bytes: the
// assignment table (rank 0 -> everyone) and the per-rank counters at the end.  One process per GPU; the ncclUniqueId travels
// over the launcher's TCP rendezvous (pretty-fast-video_amd/comm.py), the collectives run on the context's own HIP stream
// over xGMI.  librccl.so is opened at run time: single-GPU users of libpfv_hip.so need no RCCL, and the process stays free of
// a second HIP runtime (torch.distributed would bring torch's own, which doubles the host cost of small launches).
#include <dlfcn.h>

namespace pfv {

// the slice of rccl.h this file needs (ROCm 7.2 librccl.so.1; the ABI of these entry points is NCCL 2's)
struct RcclUniqueId { char internal[128]; };
typedef void *RcclComm;
enum { kRcclUint8 = 1, kRcclInt64 = 4, kRcclFloat64 = 8, kRcclSum = 0, kRcclMax = 2 };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, RcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;     // why it could not be loaded
};
static const RcclApi &rccl_api()
{
    static const RcclApi api = [] {
        RcclApi a;
        // ROCm's own build first: it is linked against the HIP runtime this library uses (a process that has also imported torch
        // holds a second librccl next to torch's bundled runtime; its streams are not ours)
        const char *names[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!a.lib) { a.why = std::string("librccl.so not found: ") + dlerror(); return a; }
        auto sym = [&](const char *n) { void *p = dlsym(a.lib, n); if (!p && a.why.empty()) a.why = std::string("librccl.so lacks ") + n; return p; };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
        a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

}  // namespace pfv

struct pfv_comm {
    pfv_ctx *ctx = nullptr;
    pfv::RcclComm comm = nullptr;
    int rank = 0, world = 1;
    double *scratch = nullptr;      // device staging for the host-value reductions
};

static int rccl_fail(pfv_ctx *ctx, int rc, const char *what)
{
    const pfv::RcclApi &a = pfv::rccl_api();
    return fail(ctx, PFV_ERR_HIP, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error"));
}

extern "C" {

PFV_API int pfv_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_comm_unique_id: null");
    const pfv::RcclApi &a = pfv::rccl_api();
    if (!a.why.empty()) return fail(nullptr, PFV_ERR_NO_DEVICE, a.why);
    pfv::RcclUniqueId id;
    const int rc = a.GetUniqueId(&id);
    if (rc) return rccl_fail(nullptr, rc, "ncclGetUniqueId");
    memcpy(id_out, id.internal, 128);
    return PFV_OK;
}

PFV_API int pfv_comm_init(pfv_ctx *ctx, int rank, int world, const uint8_t id[128], pfv_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_comm_init: bad argument");
    *out = nullptr;
    const pfv::RcclApi &a = pfv::rccl_api();
    if (!a.why.empty()) return fail(ctx, PFV_ERR_NO_DEVICE, a.why);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pfv::RcclUniqueId uid;
    memcpy(uid.internal, id, 128);
    pfv_comm *c = new pfv_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    int rc = a.CommInitRank(&c->comm, world, uid, rank);
    if (rc) { delete c; return rccl_fail(ctx, rc, "ncclCommInitRank"); }
    hipError_t e = hipMalloc((void **)&c->scratch, 64 * sizeof(double));
    if (e != hipSuccess) { a.CommDestroy(c->comm); delete c; return hip_fail(ctx, e, "pfv_comm_init"); }
    *out = c;
    return PFV_OK;
}

PFV_API int pfv_comm_rank(const pfv_comm *c) { return c ? c->rank : -1; }
PFV_API int pfv_comm_world(const pfv_comm *c) { return c ? c->world : 0; }

PFV_API int pfv_comm_broadcast_dev(pfv_comm *c, void *buf_dev, size_t bytes, int root)
{
    if (!c || !buf_dev || root < 0 || root >= c->world) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_broadcast_dev: bad argument");
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const int rc = pfv::rccl_api().Broadcast(buf_dev, buf_dev, bytes, pfv::kRcclUint8, root, c->comm, c->ctx->stream);
    return rc ? rccl_fail(c->ctx, rc, "ncclBroadcast") : PFV_OK;
}

PFV_API int pfv_comm_allreduce_f64_dev(pfv_comm *c, double *buf_dev, size_t count, int op)
{
    if (!c || !buf_dev || (op != PFV_COMM_SUM && op != PFV_COMM_MAX)) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allreduce_f64_dev: bad argument");
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const int rc = pfv::rccl_api().AllReduce(buf_dev, buf_dev, count, pfv::kRcclFloat64, op == PFV_COMM_SUM ? pfv::kRcclSum : pfv::kRcclMax, c->comm, c->ctx->stream);
    return rc ? rccl_fail(c->ctx, rc, "ncclAllReduce") : PFV_OK;
}

PFV_API int pfv_comm_allgather_dev(pfv_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank)
{
    if (!c || !send_dev || !recv_dev) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allgather_dev: bad argument");
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const int rc = pfv::rccl_api().AllGather(send_dev, recv_dev, bytes_per_rank, pfv::kRcclUint8, c->comm, c->ctx->stream);
    return rc ? rccl_fail(c->ctx, rc, "ncclAllGather") : PFV_OK;
}

// host-value convenience forms: stage through the communicator's device scratch, run the collective on the context's
// stream, synchronise, hand the result back (count <= 64)
PFV_API int pfv_comm_allreduce_f64(pfv_comm *c, double *values, size_t count, int op)
{
    if (!c || !values || count == 0 || count > 64) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allreduce_f64: bad argument");
    pfv_ctx *ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(c->scratch, values, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = pfv_comm_allreduce_f64_dev(c, c->scratch, count, op);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(values, c->scratch, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

// every rank leaves when all have arrived and the work enqueued before on each rank's stream is done (a 1-element reduction)
PFV_API int pfv_comm_barrier(pfv_comm *c)
{
    double one = 1.0;
    return pfv_comm_allreduce_f64(c, &one, 1, PFV_COMM_SUM);
}

PFV_API void pfv_comm_destroy(pfv_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) pfv::rccl_api().CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
}

}  // extern "C"
