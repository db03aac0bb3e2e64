// pfv_comm.hip -- the control-plane collectives of the multi-GPU path, straight on RCCL (included by pfv_capi.hip).
//
// The hot path shards by stream (src/enc.rs:12-26: an Encoder shares nothing with another) and by GOP (an i-frame never reads
// prev_frame, src/enc.rs:84-97): no pixel or coefficient ever crosses GPUs.  What does cross is a few hundred bytes: the
// assignment table (rank 0 -> everyone) and the per-rank counters at the end.  One process per GPU; the ncclUniqueId travels
// over the launcher's TCP rendezvous (pretty-fast-video_amd/comm.py), the collectives run on the context's own HIP stream
// over xGMI.  librccl.so is opened at run time: single-GPU users of libpfv_hip.so need no RCCL, and the process stays free of
// a second HIP runtime (torch.distributed would bring torch's own, which doubles the host cost of small launches).
#include <dlfcn.h>
#include <rccl/rccl.h>   // types, enums and prototypes only: the library itself is opened with dlopen below, nothing links against it

namespace pfv {

// the entry points this file calls, typed by the header's own prototypes
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;     // why it could not be loaded
};
static_assert(sizeof(ncclUniqueId) == 128, "pfv_comm_unique_id / pfv_comm_init carry the id as 128 bytes");
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
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *scratch = nullptr;      // device staging for the host-value reductions
};

static int rccl_fail(pfv_ctx *ctx, ncclResult_t rc, const char *what)
{
    const pfv::RcclApi &a = pfv::rccl_api();
    return fail(ctx, PFV_ERR_HIP, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(rc) : "RCCL error"));
}
// a collective enqueued while the context's stream is being captured (pfv_graph_begin) would either end up inside the graph or break
// the capture with its synchronisation: refuse
static int comm_usable(pfv_comm *c, const char *what)
{
    if (c->ctx->capturing) return fail(c->ctx, PFV_ERR_STATE, std::string(what) + ": the context is recording a graph (pfv_graph_begin)");
    return PFV_OK;
}

// also called by pfv_ctx_destroy for communicators the caller left behind
static void comm_teardown(pfv_comm *c)
{
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) pfv::rccl_api().CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
}

extern "C" {

PFV_API int pfv_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_comm_unique_id: null");
    const pfv::RcclApi &a = pfv::rccl_api();
    if (!a.why.empty()) return fail(nullptr, PFV_ERR_NO_DEVICE, a.why);
    ncclUniqueId id;
    const ncclResult_t rc = a.GetUniqueId(&id);
    if (rc != ncclSuccess) return rccl_fail(nullptr, rc, "ncclGetUniqueId");
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
    if (ctx->capturing) return fail(ctx, PFV_ERR_STATE, "pfv_comm_init: the context is recording a graph (pfv_graph_begin)");
    ncclUniqueId uid;
    memcpy(uid.internal, id, 128);
    pfv_comm *c = new pfv_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    const ncclResult_t rc = a.CommInitRank(&c->comm, world, uid, rank);
    if (rc != ncclSuccess) { delete c; return rccl_fail(ctx, rc, "ncclCommInitRank"); }
    hipError_t e = hipMalloc((void **)&c->scratch, 64 * sizeof(double));
    if (e != hipSuccess) { a.CommDestroy(c->comm); delete c; return hip_fail(ctx, e, "pfv_comm_init"); }
    { std::lock_guard<std::mutex> lk(ctx->comms_m); ctx->comms.push_back(c); }      // a communicator does not outlive its context: pfv_ctx_destroy tears down what is still here
    *out = c;
    return PFV_OK;
}

PFV_API int pfv_comm_rank(const pfv_comm *c) { return c ? c->rank : -1; }
PFV_API int pfv_comm_world(const pfv_comm *c) { return c ? c->world : 0; }

PFV_API int pfv_comm_broadcast_dev(pfv_comm *c, void *buf_dev, size_t bytes, int root)
{
    if (!c || !buf_dev || root < 0 || root >= c->world) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_broadcast_dev: bad argument");
    if (comm_usable(c, "pfv_comm_broadcast_dev")) return PFV_ERR_STATE;
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const ncclResult_t rc = pfv::rccl_api().Broadcast(buf_dev, buf_dev, bytes, ncclUint8, root, c->comm, c->ctx->stream);
    return rc != ncclSuccess ? rccl_fail(c->ctx, rc, "ncclBroadcast") : PFV_OK;
}

PFV_API int pfv_comm_allreduce_f64_dev(pfv_comm *c, double *buf_dev, size_t count, int op)
{
    if (!c || !buf_dev || (op != PFV_COMM_SUM && op != PFV_COMM_MAX)) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allreduce_f64_dev: bad argument");
    if (comm_usable(c, "pfv_comm_allreduce_f64_dev")) return PFV_ERR_STATE;
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const ncclResult_t rc = pfv::rccl_api().AllReduce(buf_dev, buf_dev, count, ncclFloat64, op == PFV_COMM_SUM ? ncclSum : ncclMax, c->comm, c->ctx->stream);
    return rc != ncclSuccess ? rccl_fail(c->ctx, rc, "ncclAllReduce") : PFV_OK;
}

PFV_API int pfv_comm_allgather_dev(pfv_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank)
{
    if (!c || !send_dev || !recv_dev) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allgather_dev: bad argument");
    if (comm_usable(c, "pfv_comm_allgather_dev")) return PFV_ERR_STATE;
    HIP_TRY(c->ctx, hipSetDevice(c->ctx->device));
    const ncclResult_t rc = pfv::rccl_api().AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream);
    return rc != ncclSuccess ? rccl_fail(c->ctx, rc, "ncclAllGather") : PFV_OK;
}

// host-value convenience forms: stage through the communicator's device scratch, run the collective on the context's
// stream, synchronise, hand the result back (count <= 64)
PFV_API int pfv_comm_allreduce_f64(pfv_comm *c, double *values, size_t count, int op)
{
    if (!c || !values || count == 0 || count > 64) return fail(c ? c->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_comm_allreduce_f64: bad argument");
    pfv_ctx *ctx = c->ctx;
    if (comm_usable(c, "pfv_comm_allreduce_f64")) return PFV_ERR_STATE;
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
    {
        std::lock_guard<std::mutex> lk(c->ctx->comms_m);
        auto &live = c->ctx->comms;
        live.erase(std::remove(live.begin(), live.end(), c), live.end());
    }
    comm_teardown(c);
}

}  // extern "C"
