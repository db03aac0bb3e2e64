// pfv_capi.hip -- the extern "C" boundary (include/pfv_hip.h) over the gfx950 kernels.
// One translation unit with the kernels: hipcc --offload-arch=gfx950 -shared.
//
// There is NO CPU fallback here: every entry point either runs the HIP kernels or returns
// a negative status.
#define PFV_CAPI_TU
#include "pfv_kernels.hip"
#ifdef PFV_SPLIT_PENC    // product build: the p-frame encode kernels are compiled on their own, with their own scheduling strategy (pfv_penc.hip)
namespace pfv {
void launch_enc_pframe_kernels(hipStream_t stream, bool flt, bool small, int compact_max, const FrameGeom &g, unsigned blocks, const uint8_t *src,
                               const uint8_t *ref, int8_t *mv, uint8_t *has, int16_t *coef, uint8_t *recon, const QTab *qt, float min_err);
}
#else
#include "pfv_penc.hip"
#endif
#include "pfv_entropy_kernels.hip"
#include "pfv_entdec_kernels.hip"
#include "pfv_synth_kernels.hip"
#include "pfv_host.hip"
#include "pfv_selfcheck.hip"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pfv_hip.h"

using namespace pfv;

// ------------------------------------------------------------------ reference constants (data)
// src/dct.rs:4-13
static const int32_t H_SCALE[64] = {
    32, 37, 34, 26, 32, 26, 34, 37, 37, 43, 39, 31, 37, 31, 39, 43, 34, 39, 35, 28, 34, 28, 35, 39, 26, 31, 28, 22, 26, 22,
    28, 31, 32, 37, 34, 26, 32, 26, 34, 37, 26, 31, 28, 22, 26, 22, 28, 31, 34, 39, 35, 28, 34, 28, 35, 39, 37, 43, 39, 31,
    37, 31, 39, 43,
};
// src/dct.rs:16-25
static const int32_t H_Q_INTRA[64] = {
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34,
    37, 40, 22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38,
    46, 56, 69, 83,
};
// src/dct.rs:28-37: all 16
static const int32_t H_Q_INTER = 16;
// src/dct.rs:39-42
static const uint8_t H_INV_ZIGZAG[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40,
    44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49,
    57, 58, 62, 63,
};

// ------------------------------------------------------------------ context
struct pfv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // grow-only device scratch for the host-pointer entry points
    void *scratch[8] = {};
    size_t scratch_cap[8] = {};
    QTab *qtab_dev = nullptr;    // 4 slots
    QTab *qtab_host = nullptr;   // pinned mirror
    int *flag_dev = nullptr;
    int n_cus = 256;             // compute units of the device (persistent-kernel grid sizing)
    bool capturing = false;      // a pfv_graph_begin is open on the stream
    int opt_enc_transform = PFV_ENC_TRANSFORM_AUTO;   // pfv_ctx_set_option(PFV_OPT_ENC_TRANSFORM)
    int opt_tile_compaction = 1;                      // pfv_ctx_set_option(PFV_OPT_TILE_COMPACTION)
    int opt_lane_mapping = PFV_LANES_AUTO;            // pfv_ctx_set_option(PFV_OPT_LANE_MAPPING)
    int opt_entropy_decode = PFV_ENTROPY_DECODE_AUTO; // pfv_ctx_set_option(PFV_OPT_ENTROPY_DECODE)
    int opt_entdec_lane_bits = (int)kEdSubBits, opt_entdec_launches = 3, opt_entdec_inner = kEdInner;   // PFV_OPT_ENTDEC_*
    std::vector<struct pfv_comm *> comms;             // live communicators on this context (pfv_comm.hip): torn down with it
    std::mutex comms_m;                               // pfv_comm_init may return on a watchdog thread (comm.py) while the main thread destroys
    pfv_ctx *owner = nullptr;                         // an object's private launch context (pfv_gop_encoder): errors are also reported on the
    //                                                   context the caller created the object on; nullptr again once that context is destroyed
    std::vector<pfv_ctx *> children;                  // the private contexts that name this one as their owner: detached by pfv_ctx_destroy
    int priority = 0;                                 // pfv_ctx_create_prio's argument (a private context inherits it)
};
// a private launch context of an object created on `user`: same device, same stream priority, errors mirrored to `user`
static int ctx_create_child(pfv_ctx *user, pfv_ctx **out)
{
    int rc = pfv_ctx_create_prio(user->device, user->priority, out);
    if (rc) return rc;
    (*out)->owner = user;
    std::lock_guard<std::mutex> lk(user->comms_m);
    user->children.push_back(*out);
    return PFV_OK;
}
static void comm_teardown(struct pfv_comm *c);

static thread_local std::string g_tls_err;

static int fail(pfv_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    if (ctx && ctx->owner) ctx->owner->err = msg;
    g_tls_err = msg;
    return code;
}
static int hip_fail(pfv_ctx *ctx, hipError_t e, const char *what)
{
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorOutOfMemory ? PFV_ERR_NOMEM : PFV_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                        \
    do {                                                          \
        hipError_t e__ = (expr);                                  \
        if (e__ != hipSuccess) return hip_fail(ctx, e__, #expr);  \
    } while (0)

static inline int pad16(int x) { return x + (16 - (x % 16)) % 16; }

extern "C" {

// PFV_BUILD_ID: hash of the sources this binary was compiled from, passed by __graft_entry__.build_hip() (hipcc -DPFV_BUILD_ID=...)
#ifndef PFV_BUILD_ID
#define PFV_BUILD_ID "unstamped"
#endif
PFV_API const char *pfv_version(void) { return "pfv-hip 0.2 (gfx950; pfv-rs 0.2.2 / codec 2.1.1 hot path; src " PFV_BUILD_ID ")"; }
PFV_API int pfv_pad16(int x) { return pad16(x); }

PFV_API const char *pfv_last_error(pfv_ctx *ctx) { return ctx ? ctx->err.c_str() : g_tls_err.c_str(); }

PFV_API int pfv_ctx_set_option(pfv_ctx *ctx, int option, int value)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    switch (option) {
    case PFV_OPT_ENC_TRANSFORM:
        if (value != PFV_ENC_TRANSFORM_AUTO && value != PFV_ENC_TRANSFORM_INT) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_ENC_TRANSFORM: unknown value");
        ctx->opt_enc_transform = value;
        return PFV_OK;
    case PFV_OPT_TILE_COMPACTION:
        if (value < 0 || value > 2) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_TILE_COMPACTION: 0, 1 or 2");
        ctx->opt_tile_compaction = value;
        return PFV_OK;
    case PFV_OPT_LANE_MAPPING:
        if (value != PFV_LANES_AUTO && value != PFV_LANES_PER_MB_8 && value != PFV_LANES_PER_MB_16) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_LANE_MAPPING: unknown value");
        ctx->opt_lane_mapping = value;
        return PFV_OK;
    case PFV_OPT_ENTROPY_DECODE:
        if (value != PFV_ENTROPY_DECODE_AUTO && value != PFV_ENTROPY_DECODE_HOST && value != PFV_ENTROPY_DECODE_DEVICE) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_ENTROPY_DECODE: unknown value");
        ctx->opt_entropy_decode = value;
        return PFV_OK;
    case PFV_OPT_ENTDEC_LANE_BITS:
        if (value < 32 || value > (int)kEdMaxSubBits || value % 32) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_ENTDEC_LANE_BITS: a multiple of 32 in 32..256");
        ctx->opt_entdec_lane_bits = value;
        return PFV_OK;
    case PFV_OPT_ENTDEC_LAUNCHES:
        if (value < 1 || value > 64) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_ENTDEC_LAUNCHES: 1..64");
        ctx->opt_entdec_launches = value;
        return PFV_OK;
    case PFV_OPT_ENTDEC_INNER_ROUNDS:
        if (value < 1 || value > 1024) return fail(ctx, PFV_ERR_BAD_ARG, "PFV_OPT_ENTDEC_INNER_ROUNDS: 1..1024");
        ctx->opt_entdec_inner = value;
        return PFV_OK;
    default:
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_ctx_set_option: unknown option");
    }
}
PFV_API int pfv_ctx_get_option(pfv_ctx *ctx, int option, int *value)
{
    if (!ctx || !value) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_ctx_get_option: bad argument");
    switch (option) {
    case PFV_OPT_ENC_TRANSFORM: *value = ctx->opt_enc_transform; return PFV_OK;
    case PFV_OPT_TILE_COMPACTION: *value = ctx->opt_tile_compaction; return PFV_OK;
    case PFV_OPT_LANE_MAPPING: *value = ctx->opt_lane_mapping; return PFV_OK;
    case PFV_OPT_ENTROPY_DECODE: *value = ctx->opt_entropy_decode; return PFV_OK;
    case PFV_OPT_ENTDEC_LANE_BITS: *value = ctx->opt_entdec_lane_bits; return PFV_OK;
    case PFV_OPT_ENTDEC_LAUNCHES: *value = ctx->opt_entdec_launches; return PFV_OK;
    case PFV_OPT_ENTDEC_INNER_ROUNDS: *value = ctx->opt_entdec_inner; return PFV_OK;
    default: return fail(ctx, PFV_ERR_BAD_ARG, "pfv_ctx_get_option: unknown option");
    }
}

PFV_API int pfv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

PFV_API int pfv_ctx_create(int device, pfv_ctx **out) { return pfv_ctx_create_prio(device, 0, out); }

PFV_API int pfv_ctx_create_prio(int device, int priority, pfv_ctx **out)
{
    if (!out) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, PFV_ERR_NO_DEVICE, "pfv_ctx_create: no HIP device visible (the HIP path has no CPU fallback)");
    }
    if (device < 0 || device >= n) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_ctx_create: device ordinal out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    pfv_ctx *ctx = new pfv_ctx();
    ctx->device = device;
    ctx->priority = priority;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
    }
    if (priority == 0) {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    } else {   // the device's greatest / least stream priority
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, priority > 0 ? greatest : least);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->qtab_dev, 4 * sizeof(QTab));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->qtab_host, 4 * sizeof(QTab), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->flag_dev, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->flag_dev, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "pfv_ctx_create");
        pfv_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return PFV_OK;
}

PFV_API void pfv_ctx_destroy(pfv_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    {   // communicators the caller did not destroy: they use this context's stream.  Their handles are INVALID from here on (pfv_hip.h)
        std::vector<pfv_comm *> live;
        { std::lock_guard<std::mutex> lk(ctx->comms_m); live.swap(ctx->comms); }
        for (pfv_comm *c : live) comm_teardown(c);
    }
    {   // objects with a private context that were created on this one and are still alive (pfv_gop_encoder): they lose their link -- no error
        // mirroring, no *_dev intake on this context's stream any more -- instead of keeping a dangling pointer
        std::lock_guard<std::mutex> lk(ctx->comms_m);
        for (pfv_ctx *c : ctx->children) c->owner = nullptr;
        ctx->children.clear();
    }
    if (ctx->owner) {   // a private context going away first: the usual order
        std::lock_guard<std::mutex> lk(ctx->owner->comms_m);
        auto &v = ctx->owner->children;
        v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
    }
    for (int i = 0; i < 8; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->qtab_dev) (void)hipFree(ctx->qtab_dev);
    if (ctx->qtab_host) (void)hipHostFree(ctx->qtab_host);
    if (ctx->flag_dev) (void)hipFree(ctx->flag_dev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// "domain:bus:device.function" of the context's device (hipDeviceGetPCIBusId): which physical GPU a rank of a sharded job sits on
PFV_API int pfv_ctx_pci_bus_id(pfv_ctx *ctx, char *out, int len)
{
    if (!ctx || !out || len < 16) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_ctx_pci_bus_id: bad argument");
    HIP_TRY(ctx, hipDeviceGetPCIBusId(out, len, ctx->device));
    return PFV_OK;
}

PFV_API int pfv_ctx_sync(pfv_ctx *ctx)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}
PFV_API void *pfv_ctx_stream(pfv_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
// every stream of the context's device (hipDeviceSynchronize), for callers that bracket a timed region
PFV_API int pfv_device_sync(pfv_ctx *ctx)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    return PFV_OK;
}

// ------------------------------------------------------------------ timing events on the context's stream
// For callers that time the kernels where they run (bench.py's roofline figure): hipEventRecord on the context's own stream
// costs a microsecond or two, a framework's event object on a foreign stream far more.
struct pfv_event {
    pfv_ctx *ctx = nullptr;
    hipEvent_t ev = nullptr;
};
PFV_API int pfv_event_create(pfv_ctx *ctx, pfv_event **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_event_create: bad argument");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev = nullptr;
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDefault));
    pfv_event *e = new pfv_event();
    e->ctx = ctx; e->ev = ev;
    *out = e;
    return PFV_OK;
}
PFV_API int pfv_event_record(pfv_event *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null event");
    HIP_TRY(e->ctx, hipEventRecord(e->ev, e->ctx->stream));
    return PFV_OK;
}
// milliseconds between two recorded events (waits for the later one)
PFV_API int pfv_event_elapsed_ms(pfv_event *start, pfv_event *stop, float *ms)
{
    if (!start || !stop || !ms) return fail(start ? start->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_event_elapsed_ms: bad argument");
    HIP_TRY(stop->ctx, hipEventSynchronize(stop->ev));
    HIP_TRY(stop->ctx, hipEventElapsedTime(ms, start->ev, stop->ev));
    return PFV_OK;
}
// the context's stream waits (on the device, not the host) for an event recorded on ANOTHER context's stream
PFV_API int pfv_ctx_wait_event(pfv_ctx *ctx, pfv_event *e)
{
    if (!ctx || !e) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_ctx_wait_event: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, e->ev, 0));
    return PFV_OK;
}
PFV_API void pfv_event_destroy(pfv_event *e)
{
    if (!e) return;
    (void)hipEventDestroy(e->ev);
    delete e;
}

// ------------------------------------------------------------------ HIP graphs over the device-pointer entry points
// One Encoder = one stream is the reference's calling pattern (src/enc.rs:125-173): 30 small launches per GOP, each of
// which costs more host time than device time for a single 1080p stream.  Every *_dev entry point only enqueues kernels on
// the context's stream, so a whole GOP can be recorded once (stream capture) and replayed as ONE graph launch.
struct pfv_graph {
    pfv_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};
PFV_API int pfv_graph_begin(pfv_ctx *ctx)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (ctx->capturing) return fail(ctx, PFV_ERR_STATE, "pfv_graph_begin: a capture is already open on this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    return PFV_OK;
}
PFV_API int pfv_graph_end(pfv_ctx *ctx, pfv_graph **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_graph_end: bad argument");
    *out = nullptr;
    if (!ctx->capturing) return fail(ctx, PFV_ERR_STATE, "pfv_graph_end: no capture is open");
    ctx->capturing = false;
    hipGraph_t graph = nullptr;
    HIP_TRY(ctx, hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return hip_fail(ctx, e, "hipGraphInstantiate");
    }
    pfv_graph *g = new pfv_graph();
    g->ctx = ctx; g->graph = graph; g->exec = exec;
    *out = g;
    return PFV_OK;
}
PFV_API int pfv_graph_launch(pfv_graph *g)
{
    if (!g) return fail(nullptr, PFV_ERR_BAD_ARG, "null graph");
    HIP_TRY(g->ctx, hipSetDevice(g->ctx->device));
    HIP_TRY(g->ctx, hipGraphLaunch(g->exec, g->ctx->stream));
    return PFV_OK;
}
PFV_API void pfv_graph_destroy(pfv_graph *g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

// Encoder::new, src/enc.rs:40-51
PFV_API int pfv_qtables_from_quality(int quality, int32_t intra_l[64], int32_t intra_c[64], int32_t inter_l[64],
                                     int32_t inter_c[64], float *px_err)
{
    if (quality < 0 || quality > 10) return fail(nullptr, PFV_ERR_BAD_ARG, "quality must be in 0..10 (src/enc.rs:38)");
    float qscale = (float)quality * 0.25f;
    if (px_err) *px_err = (float)quality * 1.5f;
    for (int i = 0; i < 64; i++) {
        if (inter_l) inter_l[i] = (int32_t)fmaxf((float)H_Q_INTER * qscale * 0.5f, 1.0f);
        if (inter_c) inter_c[i] = (int32_t)fmaxf((float)H_Q_INTER * qscale, 1.0f);
        if (intra_l) intra_l[i] = (int32_t)fmaxf((float)H_Q_INTRA[i] * qscale * 0.5f, 1.0f);
        if (intra_c) intra_c[i] = (int32_t)fmaxf((float)H_Q_INTRA[i] * qscale, 1.0f);
    }
    return PFV_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ internal helpers
// decode_only: tables read from a stream header may hold 0 (the reference's decode only multiplies, src/dct.rs:75-86);
// an encoder table must be >= 1 (it divides, src/dct.rs:95).
// QTab::rcp: fl(fl(1 / q) * (1 + 2^-21)), each step rounded to f32 (volatile: no excess precision, no contraction)
static float biased_rcp(int q)
{
    volatile float r = q ? 1.0f / (float)q : 0.0f;
    r = r * 1.000000476837158203125f;
    return r;
}
static int make_qtab(pfv_ctx *ctx, const int32_t q[64], QTab *out, bool decode_only = false)
{
    if (!q) return fail(ctx, PFV_ERR_BAD_ARG, "q-table is null");
    for (int i = 0; i < 64; i++)
        if (q[i] < (decode_only ? 0 : 1) || q[i] > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "q-table entry outside [1,65535]");
    for (int i = 0; i < 64; i++) {
        out->rcp[i] = biased_rcp(q[i]);
        int z = H_INV_ZIGZAG[i];
        out->deq[i] = (int32_t)((uint32_t)H_SCALE[z] * (uint32_t)q[z]);
    }
    return PFV_OK;
}

static int ensure_scratch(pfv_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    if (ctx->scratch_cap[slot] < bytes) {
        if (ctx->scratch[slot]) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_cap[slot] = 0;
        }
        size_t cap = (bytes + 4095) & ~(size_t)4095;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch[slot], cap));
        ctx->scratch_cap[slot] = cap;
    }
    *out = ctx->scratch[slot];
    return PFV_OK;
}

static void fill_plane(PlaneGeom &p, int w, int h, int strip0, int mb0, long src_off, long pad_off, int qsel, int clear)
{
    p.w = w; p.h = h;
    p.pw = pad16(w); p.ph = pad16(h);
    p.bw = p.pw / 16; p.bh = p.ph / 16;
    p.strips_x = (p.bw + kStripMB - 1) / kStripMB;
    p.tiles_y = (p.bh + kStripsPerWG - 1) / kStripsPerWG;
    p.strip0 = strip0; p.mb0 = mb0;
    p.tile0 = 0;
    p.qsel = qsel; p.clear = clear;
    p.src_off = src_off; p.pad_off = pad_off;
    p.fast_src = 0;
}

// geometry of a single plane handed over on its own (plane-level operators)
static FrameGeom plane_geom(int w, int h, int clear)
{
    FrameGeom g;
    memset(&g, 0, sizeof g);
    fill_plane(g.p[0], w, h, 0, 0, 0, 0, 0, clear);
    g.p[1] = g.p[0]; g.p[2] = g.p[0];
    g.n_planes = 1;
    g.strips_per_frame = g.p[0].strips_x * g.p[0].bh;
    g.tiles_per_frame = g.p[0].strips_x * g.p[0].tiles_y;
    g.mbs_per_frame = g.p[0].bw * g.p[0].bh;
    g.n_streams = 1;
    g.src_frame_bytes = (long)w * h;
    g.pad_frame_bytes = (long)g.p[0].pw * g.p[0].ph;
    g.p[0].fast_src = (w % 16 == 0);
    return g;
}

// geometry of a YUV 4:2:0 VideoFrame (src/frame.rs:12-49): chroma = (w/2) x (h/2),
// padded independently (frame.rs:31-36)
static FrameGeom frame_geom(int w, int h, int n_streams)
{
    FrameGeom g;
    memset(&g, 0, sizeof g);
    int cw = w / 2, ch = h / 2;
    fill_plane(g.p[0], w, h, 0, 0, 0, 0, 0, 0);
    long y_src = (long)w * h, y_pad = (long)g.p[0].pw * g.p[0].ph;
    int s1 = g.p[0].strips_x * g.p[0].bh, m1 = g.p[0].bw * g.p[0].bh;
    fill_plane(g.p[1], cw, ch, s1, m1, y_src, y_pad, 1, 128);
    long c_src = (long)cw * ch, c_pad = (long)g.p[1].pw * g.p[1].ph;
    int s2 = s1 + g.p[1].strips_x * g.p[1].bh, m2 = m1 + g.p[1].bw * g.p[1].bh;
    fill_plane(g.p[2], cw, ch, s2, m2, y_src + c_src, y_pad + c_pad, 1, 128);
    g.n_planes = 3;
    g.strips_per_frame = s2 + g.p[2].strips_x * g.p[2].bh;
    g.p[1].tile0 = g.p[0].strips_x * g.p[0].tiles_y;
    g.p[2].tile0 = g.p[1].tile0 + g.p[1].strips_x * g.p[1].tiles_y;
    g.tiles_per_frame = g.p[2].tile0 + g.p[2].strips_x * g.p[2].tiles_y;
    g.mbs_per_frame = m2 + g.p[2].bw * g.p[2].bh;
    g.n_streams = n_streams;
    g.src_frame_bytes = y_src + 2 * c_src;
    g.pad_frame_bytes = y_pad + 2 * c_pad;
    for (int i = 0; i < 3; i++)
        g.p[i].fast_src = (g.p[i].w % 16 == 0) && (g.p[i].src_off % 16 == 0) && (g.src_frame_bytes % 16 == 0);
    return g;
}

static FrameGeom with_base_alignment(FrameGeom g, const void *src_base)
{
    if (((uintptr_t)src_base & 15) != 0)
        for (int i = 0; i < 3; i++) g.p[i].fast_src = 0;
    return g;
}

// p-frame encoder: one workgroup per 128 x 64 tile
static inline unsigned penc_blocks(const pfv_ctx *, const FrameGeom &g)
{
    return (unsigned)((long)g.tiles_per_frame * g.n_streams);
}

// one strip per wavefront, kStripsPerWG strips per workgroup
static inline unsigned strip_blocks(const FrameGeom &g)
{
    long strips = (long)g.strips_per_frame * g.n_streams;
    return (unsigned)((strips + kStripsPerWG - 1) / kStripsPerWG);
}

// Lane mapping of the four codec kernels (pfv_kernels.hip, "Lane mappings"): 8 lanes per macroblock for launches that fill the
// device, 16 for small ones.  Measured on one MI355X (profiles/r03_lane_mappings.txt; 1080p GOP-15 encode+decode, M macroblocks/s,
// 8 vs 16 lanes): 1 stream (1 530 strips) 533 vs 567, 2 streams 687 vs 709, 4 streams (6 120 strips) 894 vs 865, one 4K stream
// (6 090 strips) 985 vs 925 -- the crossover lies between 3 060 and 6 090 strips.  PFV_OPT_LANE_MAPPING overrides the choice.
constexpr long kSmallGridStrips = 4096;
static inline bool use_small_grid(int opt, const FrameGeom &g)
{
    if (opt == PFV_LANES_PER_MB_8) return false;
    if (opt == PFV_LANES_PER_MB_16) return true;
    return (long)g.strips_per_frame * g.n_streams < kSmallGridStrips;
}
static inline unsigned half_strip_blocks(const FrameGeom &g)
{
    long waves = 2 * (long)g.strips_per_frame * g.n_streams;
    return (unsigned)((waves + kStripsPerWG - 1) / kStripsPerWG);
}
static void launch_enc_iframe(pfv_ctx *ctx, bool flt, bool small, const FrameGeom &g, const uint8_t *src, int16_t *coef, uint8_t *recon, const QTab *qt)
{
    if (small) {
        if (flt) hipLaunchKernelGGL((k_enc_iframe<true, 16>), dim3(half_strip_blocks(g)), dim3(kThreads), 0, ctx->stream, g, src, coef, recon, qt, kQuantMagic);
        else hipLaunchKernelGGL((k_enc_iframe<false, 16>), dim3(half_strip_blocks(g)), dim3(kThreads), 0, ctx->stream, g, src, coef, recon, qt, kQuantMagic);
    } else {
        if (flt) hipLaunchKernelGGL((k_enc_iframe<true, 8>), dim3(strip_blocks(g)), dim3(kThreads), 0, ctx->stream, g, src, coef, recon, qt, kQuantMagic);
        else hipLaunchKernelGGL((k_enc_iframe<false, 8>), dim3(strip_blocks(g)), dim3(kThreads), 0, ctx->stream, g, src, coef, recon, qt, kQuantMagic);
    }
}
static void launch_enc_pframe(pfv_ctx *ctx, bool flt, bool small, int compaction, const FrameGeom &g, const uint8_t *src, const uint8_t *ref, int8_t *mv,
                              uint8_t *has, int16_t *coef, uint8_t *recon, const QTab *qt, float min_err)
{
    const bool split = compaction == 2 && !small;      // PFV_OPT_TILE_COMPACTION = 2: k_pf_search + k_pf_transform
    launch_enc_pframe_kernels(ctx->stream, flt, small, split ? kPencSplit : (compaction ? kPencCompactMax : 0), g, penc_blocks(ctx, g), src, ref, mv, has, coef, recon, qt, min_err);
    if (split) {
        const unsigned tf = (unsigned)((long)g.n_streams * tf_groups_per_frame(g));
        if (flt) hipLaunchKernelGGL(k_pf_transform<true>, dim3(tf), dim3(64), 0, ctx->stream, g, src, ref, (const int8_t *)mv, (const uint8_t *)has, coef, recon, qt, kQuantMagic);
        else hipLaunchKernelGGL(k_pf_transform<false>, dim3(tf), dim3(64), 0, ctx->stream, g, src, ref, (const int8_t *)mv, (const uint8_t *)has, coef, recon, qt, kQuantMagic);
    }
}
// where a decode launch finds its coefficients: the dense [slot][macroblock][256] array, or coefficient lists (pfv_device.h: CoefLists)
struct DecCoefs {
    const int16_t *dense = nullptr;
    CoefLists lists{nullptr, nullptr};
    DecCoefs() = default;
    DecCoefs(const int16_t *d) : dense(d) {}
    DecCoefs(const uint32_t *const *entries, const uint32_t *counts) : lists{entries, counts} {}
    bool is_lists() const { return lists.entries != nullptr; }
    DecCoefs shifted(size_t slot, size_t mbs_per_frame) const
    {
        DecCoefs c;
        if (dense) c.dense = dense + slot * mbs_per_frame * 256;
        if (lists.entries) c.lists = CoefLists{lists.entries + slot, lists.counts + slot * (mbs_per_frame + 1)};
        return c;
    }
};
static void launch_dec_iframe(pfv_ctx *ctx, bool small, const FrameGeom &g, const DecCoefs &c, uint8_t *out, const QTab *qt, uint8_t *frames_out)
{
    const dim3 grid(small ? half_strip_blocks(g) : strip_blocks(g)), block(kThreads);
    if (c.is_lists()) {
        if (small) hipLaunchKernelGGL((k_dec_iframe<16, true>), grid, block, 0, ctx->stream, g, c.dense, c.lists, out, qt, frames_out);
        else hipLaunchKernelGGL((k_dec_iframe<8, true>), grid, block, 0, ctx->stream, g, c.dense, c.lists, out, qt, frames_out);
    } else {
        if (small) hipLaunchKernelGGL((k_dec_iframe<16, false>), grid, block, 0, ctx->stream, g, c.dense, c.lists, out, qt, frames_out);
        else hipLaunchKernelGGL((k_dec_iframe<8, false>), grid, block, 0, ctx->stream, g, c.dense, c.lists, out, qt, frames_out);
    }
}
static void launch_dec_pframe(pfv_ctx *ctx, bool small, const FrameGeom &g, const int8_t *mv, const uint8_t *has, const DecCoefs &c, const uint8_t *ref,
                              uint8_t *out, const QTab *qt, int *flag, uint8_t *frames_out)
{
    const dim3 grid(small ? half_strip_blocks(g) : strip_blocks(g)), block(kThreads);
    if (c.is_lists()) {
        if (small) hipLaunchKernelGGL((k_dec_pframe<16, true>), grid, block, 0, ctx->stream, g, mv, has, c.dense, c.lists, ref, out, qt, flag, frames_out);
        else hipLaunchKernelGGL((k_dec_pframe<8, true>), grid, block, 0, ctx->stream, g, mv, has, c.dense, c.lists, ref, out, qt, flag, frames_out);
    } else {
        if (small) hipLaunchKernelGGL((k_dec_pframe<16, false>), grid, block, 0, ctx->stream, g, mv, has, c.dense, c.lists, ref, out, qt, flag, frames_out);
        else hipLaunchKernelGGL((k_dec_pframe<8, false>), grid, block, 0, ctx->stream, g, mv, has, c.dense, c.lists, ref, out, qt, flag, frames_out);
    }
}

static int launch_check(pfv_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, what);
    return PFV_OK;
}

static int upload_qtabs(pfv_ctx *ctx, const int32_t *const *tables, int n)
{
    // the previous user of the pinned mirror has finished: plane-level host calls sync before returning
    for (int i = 0; i < n; i++) {
        int rc = make_qtab(ctx, tables[i], &ctx->qtab_host[i]);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->qtab_dev, ctx->qtab_host, n * sizeof(QTab), hipMemcpyHostToDevice, ctx->stream));
    return PFV_OK;
}

extern "C" {

// ------------------------------------------------------------------ plane-level operators (host buffers)
PFV_API int pfv_encode_plane(pfv_ctx *ctx, const uint8_t *px, int w, int h, const int32_t q[64], uint8_t clear,
                             int16_t *coef_out)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!px || !coef_out || w <= 0 || h <= 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_encode_plane: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameGeom g = plane_geom(w, h, clear);
    const int32_t *tabs[1] = {q};
    int rc = upload_qtabs(ctx, tabs, 1);
    if (rc) return rc;
    void *d_src, *d_coef;
    size_t coef_bytes = (size_t)g.mbs_per_frame * 512;
    if ((rc = ensure_scratch(ctx, 0, (size_t)w * h, &d_src))) return rc;
    if ((rc = ensure_scratch(ctx, 1, coef_bytes, &d_coef))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_src, px, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    // encode only: the forward transform is exact in f32 for any table
    launch_enc_iframe(ctx, ctx->opt_enc_transform != PFV_ENC_TRANSFORM_INT, use_small_grid(ctx->opt_lane_mapping, g), g, (const uint8_t *)d_src, (int16_t *)d_coef,
                      nullptr, ctx->qtab_dev);
    if ((rc = launch_check(ctx, "k_enc_iframe"))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(coef_out, d_coef, coef_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_encode_plane_delta(pfv_ctx *ctx, const uint8_t *px, int w, int h, const uint8_t *ref,
                                   const int32_t q[64], float px_err, uint8_t clear, int8_t *mv_out,
                                   uint8_t *has_coef_out, int16_t *coef_out)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!px || !ref || !mv_out || !has_coef_out || !coef_out || w <= 0 || h <= 0)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_encode_plane_delta: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameGeom g = plane_geom(w, h, clear);
    const int32_t *tabs[1] = {q};
    int rc = upload_qtabs(ctx, tabs, 1);
    if (rc) return rc;
    size_t n = (size_t)g.mbs_per_frame, coef_bytes = n * 512, pad_bytes = (size_t)g.pad_frame_bytes;
    void *d_src, *d_coef, *d_ref, *d_mv, *d_has;
    if ((rc = ensure_scratch(ctx, 0, (size_t)w * h, &d_src))) return rc;
    if ((rc = ensure_scratch(ctx, 1, coef_bytes, &d_coef))) return rc;
    if ((rc = ensure_scratch(ctx, 2, pad_bytes, &d_ref))) return rc;
    if ((rc = ensure_scratch(ctx, 3, n * 2, &d_mv))) return rc;
    if ((rc = ensure_scratch(ctx, 4, n, &d_has))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_src, px, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_ref, ref, pad_bytes, hipMemcpyHostToDevice, ctx->stream));
    float min_err = px_err * px_err * 256.0f;   // src/common.rs:209
    launch_enc_pframe(ctx, ctx->opt_enc_transform != PFV_ENC_TRANSFORM_INT, use_small_grid(ctx->opt_lane_mapping, g), ctx->opt_tile_compaction, g,
                      (const uint8_t *)d_src, (const uint8_t *)d_ref, (int8_t *)d_mv, (uint8_t *)d_has, (int16_t *)d_coef, nullptr, ctx->qtab_dev, min_err);
    if ((rc = launch_check(ctx, "k_enc_pframe"))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(coef_out, d_coef, coef_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mv_out, d_mv, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(has_coef_out, d_has, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_decode_plane_into(pfv_ctx *ctx, const int16_t *coef, int bw, int bh, const int32_t q[64],
                                  uint8_t *target)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!coef || !target || bw <= 0 || bh <= 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_decode_plane_into: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameGeom g = plane_geom(bw * 16, bh * 16, 0);
    const int32_t *tabs[1] = {q};
    int rc = upload_qtabs(ctx, tabs, 1);
    if (rc) return rc;
    size_t n = (size_t)bw * bh, coef_bytes = n * 512, pad_bytes = (size_t)g.pad_frame_bytes;
    void *d_coef, *d_out;
    if ((rc = ensure_scratch(ctx, 1, coef_bytes, &d_coef))) return rc;
    if ((rc = ensure_scratch(ctx, 5, pad_bytes, &d_out))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_coef, coef, coef_bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_dec_iframe(ctx, use_small_grid(ctx->opt_lane_mapping, g), g, (const int16_t *)d_coef, (uint8_t *)d_out, ctx->qtab_dev, nullptr);
    if ((rc = launch_check(ctx, "k_dec_iframe"))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(target, d_out, pad_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_decode_plane_delta(pfv_ctx *ctx, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef,
                                   int bw, int bh, const int32_t q[64], const uint8_t *ref, uint8_t *out)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!mv || !has_coef || !coef || !ref || !out || bw <= 0 || bh <= 0)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_decode_plane_delta: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameGeom g = plane_geom(bw * 16, bh * 16, 0);
    const int32_t *tabs[1] = {q};
    int rc = upload_qtabs(ctx, tabs, 1);
    if (rc) return rc;
    size_t n = (size_t)bw * bh, coef_bytes = n * 512, pad_bytes = (size_t)g.pad_frame_bytes;
    void *d_coef, *d_ref, *d_mv, *d_has, *d_out;
    if ((rc = ensure_scratch(ctx, 1, coef_bytes, &d_coef))) return rc;
    if ((rc = ensure_scratch(ctx, 2, pad_bytes, &d_ref))) return rc;
    if ((rc = ensure_scratch(ctx, 3, n * 2, &d_mv))) return rc;
    if ((rc = ensure_scratch(ctx, 4, n, &d_has))) return rc;
    if ((rc = ensure_scratch(ctx, 5, pad_bytes, &d_out))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_coef, coef, coef_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_ref, ref, pad_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_mv, mv, n * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_has, has_coef, n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->flag_dev, 0, sizeof(int), ctx->stream));
    launch_dec_pframe(ctx, use_small_grid(ctx->opt_lane_mapping, g), g, (const int8_t *)d_mv, (const uint8_t *)d_has, (const int16_t *)d_coef, (const uint8_t *)d_ref,
                      (uint8_t *)d_out, ctx->qtab_dev, ctx->flag_dev, nullptr);
    if ((rc = launch_check(ctx, "k_dec_pframe"))) return rc;
    int flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->flag_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flag) return fail(ctx, PFV_ERR_BAD_MV, "motion vector points outside the reference plane (src/common.rs:258-259)");
    HIP_TRY(ctx, hipMemcpy(out, d_out, pad_bytes, hipMemcpyDeviceToHost));
    return PFV_OK;
}

PFV_API int pfv_decode_plane_delta_into(pfv_ctx *ctx, const int8_t *mv, const uint8_t *has_coef,
                                        const int16_t *coef, int bw, int bh, const int32_t q[64],
                                        uint8_t *ref_and_target)
{
    // read-all-then-write-all (src/common.rs:498-521): the device reads plane A and writes plane B
    return pfv_decode_plane_delta(ctx, mv, has_coef, coef, bw, bh, q, ref_and_target, ref_and_target);
}

PFV_API int pfv_blit_dev(pfv_ctx *ctx, uint8_t *dst, int dst_w, int dst_h, const uint8_t *src, int src_w, int src_h,
                         int dx, int dy, int sx, int sy, int sw, int sh)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!dst || !src || sw < 0 || sh < 0 || dx < 0 || dy < 0 || sx < 0 || sy < 0 || dx + sw > dst_w || dy + sh > dst_h ||
        sx + sw > src_w || sy + sh > src_h)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_blit_dev: rectangle outside a plane (the reference panics on slice bounds)");
    if (sw == 0 || sh == 0) return PFV_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long n = (long)sw * sh;
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_blit, dim3(blocks), dim3(kThreads), 0, ctx->stream, dst, dst_w, src, src_w, dx, dy, sx, sy, sw, sh);
    return launch_check(ctx, "k_blit");
}

// VideoPlane::reduce / VideoPlane::double (src/common.rs:523-556) on device-resident planes
PFV_API int pfv_reduce_dev(pfv_ctx *ctx, uint8_t *dst, const uint8_t *src, int src_w, int src_h)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!dst || !src || src_w < 0 || src_h < 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_reduce_dev: bad argument");
    long n = (long)(src_w / 2) * (src_h / 2);
    if (n == 0) return PFV_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_reduce2x, dim3(blocks), dim3(kThreads), 0, ctx->stream, dst, src, src_w, src_h);
    return launch_check(ctx, "k_reduce2x");
}
PFV_API int pfv_double_dev(pfv_ctx *ctx, uint8_t *dst, const uint8_t *src, int src_w, int src_h)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!dst || !src || src_w < 0 || src_h < 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_double_dev: bad argument");
    long n = (long)src_w * src_h * 4;
    if (n == 0) return PFV_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int blocks = (int)((n + kThreads - 1) / kThreads);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_double2x, dim3(blocks), dim3(kThreads), 0, ctx->stream, dst, src, src_w, src_h);
    return launch_check(ctx, "k_double2x");
}

// RGB8 <-> planar YUV 4:2:0 frames, the conversions of the reference's test helpers (src/lib.rs:337-394)
PFV_API int pfv_rgb_to_yuv420_dev(pfv_ctx *ctx, const uint8_t *rgb_dev, int width, int height, uint8_t *frame_dev)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!rgb_dev || !frame_dev || width <= 0 || height <= 0 || (width & 1) || (height & 1))
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_rgb_to_yuv420_dev: null buffer or odd / non-positive size (src/frame.rs:13)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long n = (long)width * height;
    int blocks = (int)std::min<long>((n + kThreads - 1) / kThreads, 8192);
    hipLaunchKernelGGL(k_rgb_to_yuv420, dim3(blocks), dim3(kThreads), 0, ctx->stream, rgb_dev, width, height, frame_dev);
    return launch_check(ctx, "k_rgb_to_yuv420");
}
PFV_API int pfv_yuv420_to_rgb_dev(pfv_ctx *ctx, const uint8_t *frame_dev, int width, int height, uint8_t *rgb_dev)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!rgb_dev || !frame_dev || width <= 0 || height <= 0 || (width & 1) || (height & 1))
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_yuv420_to_rgb_dev: null buffer or odd / non-positive size (src/frame.rs:13)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    long n = (long)width * height;
    int blocks = (int)std::min<long>((n + kThreads - 1) / kThreads, 8192);
    hipLaunchKernelGGL(k_yuv420_to_rgb, dim3(blocks), dim3(kThreads), 0, ctx->stream, frame_dev, width, height, rgb_dev);
    return launch_check(ctx, "k_yuv420_to_rgb");
}

// ------------------------------------------------------------------ device memory helpers
// Synthetic workload generator: frame `t` of n_streams streams, stream s seeded with seeds[s], written as
// packed Y|U|V frames back to back into frames_dev.  Same bytes as synth.SyntheticStream(width, height, seed).frame(t).
PFV_API int pfv_synth_frames_dev(pfv_ctx *ctx, int width, int height, int n_streams, const uint64_t *seeds, int t, uint8_t *frames_dev)
{
    return pfv_synth_frames_kind_dev(ctx, width, height, n_streams, seeds, t, PFV_SYNTH_PAN, frames_dev);
}
PFV_API int pfv_synth_frames_kind_dev(pfv_ctx *ctx, int width, int height, int n_streams, const uint64_t *seeds, int t, int kind,
                                      uint8_t *frames_dev)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!seeds || !frames_dev || width <= 0 || height <= 0 || (width & 1) || (height & 1) || n_streams <= 0 || n_streams > 65535 || t < 0 ||
        (kind != PFV_SYNTH_PAN && kind != PFV_SYNTH_LOW_MOTION && kind != PFV_SYNTH_STATIC))
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_synth_frames_dev: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *sd = nullptr;
    int rc = ensure_scratch(ctx, 7, (size_t)n_streams * sizeof(uint64_t), &sd);
    if (rc) return rc;
    // pageable source: the runtime stages the few bytes before returning, the caller's array is free again
    HIP_TRY(ctx, hipMemcpyAsync(sd, seeds, (size_t)n_streams * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    const long n = (long)width * height;
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads), 3, (unsigned)n_streams);
    if (kind != PFV_SYNTH_PAN)
        hipLaunchKernelGGL(k_synth_frames_low_motion, grid, dim3(kThreads), 0, ctx->stream, width, height, t, (const uint64_t *)sd, frames_dev,
                           (long)pfv_frame_bytes(width, height), kind == PFV_SYNTH_LOW_MOTION ? kSynthObjects : 0);
    else
        hipLaunchKernelGGL(k_synth_frames, grid, dim3(kThreads), 0, ctx->stream, width, height, t, (const uint64_t *)sd, frames_dev,
                           (long)pfv_frame_bytes(width, height));
    return launch_check(ctx, "k_synth_frames");
}

PFV_API int pfv_dev_alloc(pfv_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dev_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(out, bytes ? bytes : 16));
    return PFV_OK;
}
PFV_API int pfv_dev_free(pfv_ctx *ctx, void *p)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!p) return PFV_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(p));
    return PFV_OK;
}
// Page-locked host memory for the host-buffer entry points: copies from / to it run at PCIe rate.
PFV_API int pfv_host_alloc(pfv_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out || !bytes) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_host_alloc: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return PFV_OK;
}
PFV_API int pfv_host_free(pfv_ctx *ctx, void *p)
{
    if (!ctx) return fail(nullptr, PFV_ERR_BAD_ARG, "null ctx");
    if (!p) return PFV_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostFree(p));
    return PFV_OK;
}
PFV_API int pfv_dev_upload(pfv_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!ctx || !dst_dev || !src_host) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dev_upload: bad argument");
    HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}
PFV_API int pfv_dev_download(pfv_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (!ctx || !dst_host || !src_dev) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dev_download: bad argument");
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

// device-to-device copy on the context's stream (asynchronous: ordered like every *_dev call) -- e.g. a consumer that keeps a frame a decoder
// left in device memory beyond the call that hands it over
PFV_API int pfv_dev_copy(pfv_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    if (!ctx || !dst_dev || !src_dev) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dev_copy: bad argument");
    HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PFV_OK;
}

// ------------------------------------------------------------------ frame geometry queries
PFV_API size_t pfv_frame_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * height + 2 * (size_t)(width / 2) * (height / 2);
}
PFV_API size_t pfv_padded_frame_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)pad16(width) * pad16(height) + 2 * (size_t)pad16(width / 2) * pad16(height / 2);
}
PFV_API int pfv_total_blocks(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (pad16(width) / 16) * (pad16(height) / 16) + 2 * (pad16(width / 2) / 16) * (pad16(height / 2) / 16);
}

}  // extern "C"

// ------------------------------------------------------------------ sessions
static int init_padded(pfv_ctx *ctx, const FrameGeom &g, uint8_t *buf)
{
    dim3 grid(64, 3, g.n_streams);
    hipLaunchKernelGGL(k_init_padded, grid, dim3(kThreads), 0, ctx->stream, g, buf);
    return launch_check(ctx, "k_init_padded");
}

// ------------------------------------------------------------------ may the encoder run its transforms in f32?
// The float kernels (k_enc_*<true>) are exact as long as every intermediate is an integer below 2^24 (pfv_kernels.hip, "the
// same transforms in f32").  The forward transform is: |fdct2d| <= 128 * 256 * (row norm)^2 = 2.5 M for any 8-bit input.  For
// the closed-loop inverse the bound depends on the tables: with M(u,v) = that forward bound, the largest coefficient is
// floor(floor(M * SCALE / 65536) / q), decode multiplies it by SCALE[z] * q[z] at its zigzag position z (src/dct.rs:78-82), and an L1
// bound pushes all 64 such maxima through |idct| columns and rows at once (with slack for the truncations).  Quality-derived
// tables give 1.9 M; a table for which the bound reaches 2^23 keeps the integer kernels.
// |d out / d in| of the two 1-D transforms (the integer butterflies on scaled unit vectors) and the signs of those derivatives;
// built once, thread-safely (function-local static), sessions may be created from several threads
struct XformNorms {
    double F1[8];            // L1 norm of each forward output
    double Iabs[8][8];       // |inverse|
    signed char fsign[64];   // [u * 8 + k]: sign of d fdct(out u) / d (in k)
    signed char isign[64];
};
static const XformNorms &xform_norms()
{
    static const XformNorms t = [] {
        XformNorms n{};
        for (int k = 0; k < 8; k++) {
            int f[8] = {0, 0, 0, 0, 0, 0, 0, 0}, i8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            f[k] = i8[k] = 1 << 20;
            fdct8(f);
            idct8(i8);
            for (int u = 0; u < 8; u++) {
                n.F1[u] += fabs((double)f[u]) / (1 << 20);
                n.Iabs[u][k] = fabs((double)i8[u]) / (1 << 20);
                n.fsign[u * 8 + k] = (signed char)(f[u] < 0 ? -1 : 1);
                n.isign[u * 8 + k] = (signed char)(i8[u] < 0 ? -1 : 1);
            }
        }
        return n;
    }();
    return t;
}
// largest coefficient magnitude the encoder can produce at raster position i for inputs of the given amplitude (24.8 fixed point)
static double enc_max_coef(const int32_t q[64], double amplitude, int i)
{
    const XformNorms &n = xform_norms();
    const double M = amplitude * n.F1[i >> 3] * n.F1[i & 7];
    return floor(floor(M * H_SCALE[i] / 65536.0) / (double)q[i]);
}
static bool enc_float_exact(const int32_t q[64], double amplitude)
{
    const XformNorms &nm = xform_norms();
    double D[8][8], worst = 0;
    for (int i = 0; i < 64; i++) {
        const int u = i >> 3, v = i & 7, z = H_INV_ZIGZAG[i];
        const double M = amplitude * nm.F1[u] * nm.F1[v];
        // decode puts encode's coefficient of raster i back at raster i (slot z = INV_ZIGZAG[i] is where encode stored it) but
        // multiplies it by the table entries at index z (src/dct.rs:78-82)
        const double c = enc_max_coef(q, amplitude, i);
        D[u][v] = c * (double)H_SCALE[z] * (double)q[z];
        worst = std::max(worst, std::max(M, D[u][v]));
    }
    double col[8][8];
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            double a = 16;
            for (int k = 0; k < 8; k++) a += nm.Iabs[u][k] * D[k][v];     // columns first (src/common.rs:315)
            col[u][v] = a;
            worst = std::max(worst, a);
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            double a = 16;
            for (int k = 0; k < 8; k++) a += col[u][k] * nm.Iabs[v][k];   // then rows
            worst = std::max(worst, a);
        }
    return worst < 8388608.0;   // 2^23: a factor 2 below what f32 holds exactly
}

struct pfv_enc_session {
    pfv_ctx *ctx = nullptr;
    int width = 0, height = 0, n_streams = 0;
    FrameGeom geom;
    QTab *qtab_dev = nullptr;       // intra_l, intra_c, inter_l, inter_c
    float px_err = 0.0f;
    bool flt = false;                        // the closed loop may run in f32 (enc_float_exact holds for all four tables)
    int tile_compaction = 1;                 // PFV_OPT_TILE_COMPACTION at creation
    int lane_mapping = PFV_LANES_AUTO;       // PFV_OPT_LANE_MAPPING at creation
    uint8_t *prev[2] = {nullptr, nullptr};   // ping-pong prev_frame, padded, n_streams wide
    int cur = 0;                             // prev[cur] is the current prev_frame
    int win_first = 0, win_count = 0;        // slot window of the *_dev / pack calls (pfv_enc_session_set_window)
    size_t in_stride = 0;                    // bytes between the input frames of consecutive slots (0: packed)
    // staging for the host-buffer entry points
    uint8_t *st_frames = nullptr;
    int16_t *st_coef = nullptr;
    int8_t *st_mv = nullptr;
    uint8_t *st_has = nullptr;
    // device entropy stage (pfv_enc_entropy_enable)
    bool ent_on = false;
    uint32_t ent_cap = 0;
    EntBufs ent{};
    std::vector<void *> ent_allocs;
    std::vector<uint32_t> ent_sizes;         // last pfv_enc_payload_sizes result
    // optional second HIP stream for the stage (pfv_enc_entropy_set_async): the memory-bound k_ent_* kernels of frame t
    // overlap the VALU-bound encode kernel of frame t+1
    uint8_t *ent_packed = nullptr;           // all payloads back to back (pfv_enc_payloads_fetch)
    uint32_t *ent_offsets_dev = nullptr;
    size_t ent_packed_cap = 0;
    hipStream_t ent_stream = nullptr;
    hipEvent_t ev_encoded = nullptr;         // main stream: the buffers handed to pack are complete
    hipEvent_t ev_packed[2] = {nullptr, nullptr};   // entropy stream: pack call t has finished with its inputs
    int ev_cur = 0;
    bool ev_prev_valid = false;
};

struct pfv_dec_session {
    pfv_ctx *ctx = nullptr;
    int width = 0, height = 0, n_streams = 0, n_qtables = 0;
    FrameGeom geom;
    QTab *qtab_dev = nullptr;
    uint8_t *fb[2] = {nullptr, nullptr};     // ping-pong framebuffer
    int lane_mapping = PFV_LANES_AUTO;       // PFV_OPT_LANE_MAPPING at creation
    int cur = 0;
    int *flag_dev = nullptr;                 // [n_streams]: a p-frame decode met a motion vector that leaves the plane
    std::vector<int> flags_host;
    uint8_t *frames_out = nullptr;           // optional fused retframe output (pfv_dec_set_output_dev)
    size_t out_stride = 0;                   // bytes between the output frames of consecutive slots (0: packed)
    int win_first = 0, win_count = 0;        // slot window of the *_dev calls (pfv_dec_session_set_window)
    int16_t *st_coef = nullptr;
    int8_t *st_mv = nullptr;
    uint8_t *st_has = nullptr;
    uint8_t *st_frames = nullptr;
    uint32_t *st_idx = nullptr;              // sparse coefficient upload (pfv_dec_*_sparse)
    int16_t *st_val = nullptr;
    size_t st_sparse_cap = 0;
};

extern "C" {

PFV_API int pfv_enc_session_create(pfv_ctx *ctx, int width, int height, int quality, int n_streams,
                                   pfv_enc_session **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_session_create: bad argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (width & 1) || (height & 1) || width > 65535 || height > 65535)
        return fail(ctx, PFV_ERR_BAD_ARG, "width/height must be even (src/frame.rs:13) and fit u16 (src/enc.rs:195-196)");
    if (quality < 0 || quality > 10) return fail(ctx, PFV_ERR_BAD_ARG, "quality must be in 0..10 (src/enc.rs:38)");
    if (n_streams <= 0) return fail(ctx, PFV_ERR_BAD_ARG, "n_streams must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pfv_enc_session *s = new pfv_enc_session();
    s->ctx = ctx; s->width = width; s->height = height; s->n_streams = n_streams;
    s->win_count = n_streams;
    s->geom = frame_geom(width, height, n_streams);
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], &s->px_err);
    QTab tabs[4];
    for (int i = 0; i < 4; i++) {
        int rc = make_qtab(ctx, q[i], &tabs[i]);
        if (rc) { delete s; return rc; }
    }
    s->tile_compaction = ctx->opt_tile_compaction;
    s->lane_mapping = ctx->opt_lane_mapping;
    s->flt = ctx->opt_enc_transform != PFV_ENC_TRANSFORM_INT && enc_float_exact(q[0], 128.0 * 256.0) && enc_float_exact(q[1], 128.0 * 256.0) &&
             enc_float_exact(q[2], 127.0 * 256.0) && enc_float_exact(q[3], 127.0 * 256.0);
    size_t pad_bytes = (size_t)s->geom.pad_frame_bytes * n_streams;
    hipError_t e = hipMalloc((void **)&s->qtab_dev, sizeof tabs);
    if (e == hipSuccess) e = hipMemcpy(s->qtab_dev, tabs, sizeof tabs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&s->prev[0], pad_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->prev[1], pad_bytes);
    if (e != hipSuccess) {
        int rc = hip_fail(ctx, e, "pfv_enc_session_create");
        pfv_enc_session_destroy(s);
        return rc;
    }
    // prev_frame = VideoFrame::new_padded (src/enc.rs:46)
    int rc = init_padded(ctx, s->geom, s->prev[0]);
    if (!rc) rc = init_padded(ctx, s->geom, s->prev[1]);
    if (rc) { pfv_enc_session_destroy(s); return rc; }
    *out = s;
    return PFV_OK;
}

PFV_API void pfv_enc_session_destroy(pfv_enc_session *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    void *bufs[] = {s->qtab_dev, s->prev[0], s->prev[1], s->st_frames, s->st_coef, s->st_mv, s->st_has};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (void *b : s->ent_allocs)
        if (b) (void)hipFree(b);
    if (s->ent_packed) (void)hipFree(s->ent_packed);
    if (s->ent_offsets_dev) (void)hipFree(s->ent_offsets_dev);
    if (s->ent_stream) {
        (void)hipStreamSynchronize(s->ent_stream);
        (void)hipEventDestroy(s->ev_encoded);
        (void)hipEventDestroy(s->ev_packed[0]);
        (void)hipEventDestroy(s->ev_packed[1]);
        (void)hipStreamDestroy(s->ent_stream);
    }
    delete s;
}

// Slots [first, first + count) of a session: everything the kernels index by stream is a base pointer + stream x stride, so a
// window is the same launch on shifted base pointers with n_streams = count.  Does not touch the ping-pong index: a frame step may
// consist of several windows (pfv_gop_encoder: the GOPs still running at step t need not be neighbours).
static FrameGeom enc_win_geom(const pfv_enc_session *s, int count, const uint8_t *frames_win)
{
    FrameGeom g = s->geom;
    g.n_streams = count;
    if (s->in_stride) {
        g.src_frame_bytes = (long)s->in_stride;
        if (s->in_stride % 16)
            for (int i = 0; i < 3; i++) g.p[i].fast_src = 0;
    }
    return with_base_alignment(g, frames_win);
}
static int enc_launch(pfv_enc_session *s, bool pframe, int first, int count, const uint8_t *frames_dev, int8_t *mv_dev, uint8_t *has_dev,
                      int16_t *coef_dev)
{
    pfv_ctx *ctx = s->ctx;
    const size_t stride = s->in_stride ? s->in_stride : (size_t)s->geom.src_frame_bytes;
    const uint8_t *src = frames_dev + (size_t)first * stride;
    const size_t mb0 = (size_t)first * (size_t)s->geom.mbs_per_frame, pad0 = (size_t)first * (size_t)s->geom.pad_frame_bytes;
    const FrameGeom g = enc_win_geom(s, count, src);
    const int nxt = s->cur ^ 1;
    if (pframe) {
        const float min_err = s->px_err * s->px_err * 256.0f;   // src/common.rs:209
        launch_enc_pframe(ctx, s->flt, use_small_grid(s->lane_mapping, g), s->tile_compaction, g, src, s->prev[s->cur] + pad0, mv_dev + mb0 * 2,
                          has_dev + mb0, coef_dev + mb0 * 256, s->prev[nxt] + pad0, s->qtab_dev + 2, min_err);
        return launch_check(ctx, "k_enc_pframe");
    }
    launch_enc_iframe(ctx, s->flt, use_small_grid(s->lane_mapping, g), g, src, coef_dev + mb0 * 256, s->prev[nxt] + pad0, s->qtab_dev + 0);
    return launch_check(ctx, "k_enc_iframe");
}

PFV_API int pfv_enc_session_set_window(pfv_enc_session *s, int first, int count)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (first < 0 || count <= 0 || first > s->n_streams - count) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_enc_session_set_window: window outside [0, n_streams)");
    s->win_first = first; s->win_count = count;
    return PFV_OK;
}
PFV_API int pfv_enc_session_set_frame_stride(pfv_enc_session *s, size_t bytes)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (bytes && bytes < (size_t)s->geom.src_frame_bytes) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_enc_session_set_frame_stride: stride below pfv_frame_bytes");
    s->in_stride = bytes;
    return PFV_OK;
}
static bool enc_full_window(const pfv_enc_session *s) { return s->win_first == 0 && s->win_count == s->n_streams && s->in_stride == 0; }

PFV_API int pfv_enc_iframe_dev(pfv_enc_session *s, const uint8_t *frames_dev, int16_t *coef_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames_dev || !coef_dev) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_iframe_dev: null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_launch(s, false, s->win_first, s->win_count, frames_dev, nullptr, nullptr, coef_dev);
    if (rc) return rc;
    s->cur ^= 1;
    return PFV_OK;
}

PFV_API int pfv_enc_pframe_dev(pfv_enc_session *s, const uint8_t *frames_dev, int8_t *mv_dev, uint8_t *has_coef_dev,
                               int16_t *coef_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames_dev || !mv_dev || !has_coef_dev || !coef_dev) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_pframe_dev: null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_launch(s, true, s->win_first, s->win_count, frames_dev, mv_dev, has_coef_dev, coef_dev);
    if (rc) return rc;
    s->cur ^= 1;
    return PFV_OK;
}

static int enc_staging(pfv_enc_session *s)
{
    pfv_ctx *ctx = s->ctx;
    if (s->st_frames) return PFV_OK;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMalloc((void **)&s->st_frames, (size_t)s->geom.src_frame_bytes * s->n_streams));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_coef, n * 512));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_mv, n * 2));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_has, n));
    return PFV_OK;
}

PFV_API int pfv_enc_iframe(pfv_enc_session *s, const uint8_t *frames, int16_t *coef_out)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames || !coef_out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_iframe: null buffer");
    if (!enc_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_enc_iframe: the host-buffer entry points work on all slots, packed (reset the window / frame stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_staging(s);
    if (rc) return rc;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMemcpyAsync(s->st_frames, frames, (size_t)s->geom.src_frame_bytes * s->n_streams,
                                hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pfv_enc_iframe_dev(s, s->st_frames, s->st_coef))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(coef_out, s->st_coef, n * 512, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_enc_pframe(pfv_enc_session *s, const uint8_t *frames, int8_t *mv_out, uint8_t *has_coef_out,
                           int16_t *coef_out)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames || !mv_out || !has_coef_out || !coef_out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_pframe: null buffer");
    if (!enc_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_enc_pframe: the host-buffer entry points work on all slots, packed (reset the window / frame stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_staging(s);
    if (rc) return rc;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMemcpyAsync(s->st_frames, frames, (size_t)s->geom.src_frame_bytes * s->n_streams,
                                hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pfv_enc_pframe_dev(s, s->st_frames, s->st_mv, s->st_has, s->st_coef))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(coef_out, s->st_coef, n * 512, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mv_out, s->st_mv, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(has_coef_out, s->st_has, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API const uint8_t *pfv_enc_prev_frame_dev(pfv_enc_session *s, int stream)
{
    if (!s || stream < 0 || stream >= s->n_streams) return nullptr;
    return s->prev[s->cur] + (size_t)stream * s->geom.pad_frame_bytes;
}

PFV_API int pfv_enc_prev_frame(pfv_enc_session *s, uint8_t *out_host)
{
    if (!s || !out_host) return fail(s ? s->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_enc_prev_frame: bad argument");
    pfv_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, s->prev[s->cur], (size_t)s->geom.pad_frame_bytes * s->n_streams,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

// ------------------------------------------------------------------ device entropy stage of the encoder session
// Packet payloads (enc.rs:237-320, :332-470) built on the device from the buffers the encode entry points produced:
// byte-identical to serialize_iframe / serialize_pframe (pfv_host.hip) on the same coefficients.
PFV_API size_t pfv_payload_worst_case(int width, int height)
{
    // 19 header bytes + per macroblock a 16-bit block header and 256 x (two 15-bit codes + 15 value bits)
    size_t tb = (size_t)pfv_total_blocks(width, height);
    size_t bits = 19 * 8 + tb * 16 + tb * 256 * 45;
    return ((bits + 7) / 8 + 3) & ~(size_t)3;
}

PFV_API int pfv_enc_entropy_enable(pfv_enc_session *s, size_t payload_cap)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (s->ent_on) return PFV_OK;
    size_t cap = payload_cap ? payload_cap : pfv_payload_worst_case(s->width, s->height);
    cap = (cap + 15) & ~(size_t)15;   // 16-byte stride: k_ent_gather moves uint4s
    if (cap < 24 || cap > 0xfffffff0u) return fail(ctx, PFV_ERR_BAD_ARG, "payload capacity must be in [24, 2^32)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t S = (size_t)s->n_streams, tb = (size_t)s->geom.mbs_per_frame, n_sb = tb * 4;
    auto grab = [&](void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) s->ent_allocs.push_back(*p);
        return e;
    };
    const size_t n_groups = (n_sb + kEntThreads - 1) / kEntThreads;
    hipError_t e = grab((void **)&s->ent.syms, S * n_groups * kEntGroupSyms * 4);
    if (e == hipSuccess) e = grab((void **)&s->ent.groups, S * n_groups * sizeof(EntGroup));
    if (e == hipSuccess) e = grab((void **)&s->ent.hist, S * 16 * 4);
    if (e == hipSuccess) e = grab((void **)&s->ent.codes, S * sizeof(EntCodes));
    if (e == hipSuccess) e = grab((void **)&s->ent.sizes, S * 4);
    if (e == hipSuccess) e = grab((void **)&s->ent.payload, S * cap);
    if (e == hipSuccess) e = hipMemsetAsync(s->ent.hist, 0, S * 16 * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->ent.codes, 0, S * sizeof(EntCodes), ctx->stream);
    if (e != hipSuccess) {
        for (void *b : s->ent_allocs) (void)hipFree(b);
        s->ent_allocs.clear();
        s->ent = EntBufs{};
        return hip_fail(ctx, e, "pfv_enc_entropy_enable");
    }
    s->ent_cap = (uint32_t)cap;
    s->ent_sizes.assign(S, 0);
    s->ent_on = true;
    return PFV_OK;
}

// slots [first, first + count): every buffer of the stage is indexed by stream, so a window is the same launches on shifted bases
static int ent_pack_win(pfv_enc_session *s, bool pframe, int first, int count, const int8_t *mv_dev, const uint8_t *has_dev, const int16_t *coef_dev)
{
    pfv_ctx *ctx = s->ctx;
    if (!s->ent_on) return fail(ctx, PFV_ERR_STATE, "call pfv_enc_entropy_enable first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    EntFrame f{};
    f.total_blocks = s->geom.mbs_per_frame;
    f.n_streams = count;
    f.n_groups = (f.total_blocks * 4 + kEntThreads - 1) / kEntThreads;
    f.pframe = pframe ? 1 : 0;
    f.cap_bytes = s->ent_cap;
    f.ones16 = 0x00010001u;
    f.qidx[0] = pframe ? 2 : 0;                    // intra_l, intra_c, intra_c / inter_l, inter_c, inter_c
    f.qidx[1] = f.qidx[2] = pframe ? 3 : 1;        // (enc.rs:296-298, :409-411)
    const size_t k = (size_t)first, tb = (size_t)f.total_blocks, ng = (size_t)f.n_groups;
    EntBufs b = s->ent;
    b.coef = coef_dev + k * tb * 256;
    b.mv = mv_dev ? mv_dev + k * tb * 2 : nullptr;
    b.has = has_dev ? has_dev + k * tb : nullptr;
    b.syms += k * ng * kEntGroupSyms; b.groups += k * ng; b.hist += k * 16; b.codes += k; b.sizes += k;
    b.payload += k * (size_t)s->ent_cap;
    const dim3 per_sb((unsigned)f.n_groups, (unsigned)f.n_streams);
    hipStream_t st = ctx->stream;
    if (s->ent_stream) {   // inputs are complete once the main stream reaches this point
        st = s->ent_stream;
        HIP_TRY(ctx, hipEventRecord(s->ev_encoded, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(st, s->ev_encoded, 0));
    }
    hipLaunchKernelGGL(k_ent_scan, per_sb, dim3(kEntThreads), 0, st, f, b);
    hipLaunchKernelGGL(k_ent_codes, dim3((unsigned)f.n_streams), dim3(kEntThreads), 0, st, f, b);
    hipLaunchKernelGGL(k_ent_init, dim3(64, (unsigned)f.n_streams), dim3(kEntThreads), 0, st, f, b);
    hipLaunchKernelGGL(k_ent_pack, per_sb, dim3(kEntThreads), 0, st, f, b);
    int rc = launch_check(ctx, "k_ent_*");
    if (rc || !s->ent_stream) return rc;
    // The caller alternates between two sets of coefficient / header buffers: the encode call after this one writes the
    // other set and may overlap this stage; the one after that reuses this set, so the main stream waits here for the
    // PREVIOUS pack call -- everything enqueued on it later is ordered behind that call's reads.
    HIP_TRY(ctx, hipEventRecord(s->ev_packed[s->ev_cur], st));
    if (s->ev_prev_valid) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->ev_packed[s->ev_cur ^ 1], 0));
    s->ev_cur ^= 1;
    s->ev_prev_valid = true;
    return PFV_OK;
}
static int ent_pack(pfv_enc_session *s, bool pframe, const int8_t *mv_dev, const uint8_t *has_dev, const int16_t *coef_dev)
{
    return ent_pack_win(s, pframe, s->win_first, s->win_count, mv_dev, has_dev, coef_dev);
}
// Runs the stage on its own HIP stream (1) or on the context's stream (0, default).  With 1 the caller must alternate
// between TWO sets of device buffers for the encode outputs it packs; pfv_enc_payload_sizes / _fetch synchronise with the
// stage, pfv_enc_entropy_join makes the context's stream wait for it without blocking the host.
PFV_API int pfv_enc_entropy_set_async(pfv_enc_session *s, int on)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !s->ent_stream) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&s->ent_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_encoded, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_packed[0], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_packed[1], hipEventDisableTiming));
        s->ev_prev_valid = false;
    } else if (!on && s->ent_stream) {
        HIP_TRY(ctx, hipStreamSynchronize(s->ent_stream));
        (void)hipEventDestroy(s->ev_encoded);
        (void)hipEventDestroy(s->ev_packed[0]);
        (void)hipEventDestroy(s->ev_packed[1]);
        (void)hipStreamDestroy(s->ent_stream);
        s->ent_stream = nullptr;
    }
    return PFV_OK;
}
PFV_API int pfv_enc_entropy_join(pfv_enc_session *s)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!s->ent_stream || !s->ev_prev_valid) return PFV_OK;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->ev_packed[s->ev_cur ^ 1], 0));
    return PFV_OK;
}
PFV_API int pfv_enc_pack_iframe_dev(pfv_enc_session *s, const int16_t *coef_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!coef_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_enc_pack_iframe_dev: null buffer");
    return ent_pack(s, false, nullptr, nullptr, coef_dev);
}
PFV_API int pfv_enc_pack_pframe_dev(pfv_enc_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev, const int16_t *coef_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!mv_dev || !has_coef_dev || !coef_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_enc_pack_pframe_dev: null buffer");
    return ent_pack(s, true, mv_dev, has_coef_dev, coef_dev);
}
// Payload byte counts of the last pack call, one per stream (synchronises the context's stream).  PFV_ERR_FORMAT when a
// coefficient needs more than 15 size bits (the reference panics in rle.rs:44), PFV_ERR_NOMEM when a payload exceeds
// the capacity; `sizes_out` is filled either way (failed streams read 0).
PFV_API int pfv_enc_payload_sizes(pfv_enc_session *s, uint32_t *sizes_out)
{
    if (!s || !sizes_out) return fail(s ? s->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_enc_payload_sizes: bad argument");
    pfv_ctx *ctx = s->ctx;
    if (!s->ent_on) return fail(ctx, PFV_ERR_STATE, "call pfv_enc_entropy_enable first");
    hipStream_t st = s->ent_stream ? s->ent_stream : ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(s->ent_sizes.data(), s->ent.sizes, (size_t)s->n_streams * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int rc = PFV_OK;
    for (int i = 0; i < s->n_streams; i++) {
        uint32_t v = s->ent_sizes[i];
        if (v == kEntErrOversize) { rc = PFV_ERR_FORMAT; v = 0; }
        else if (v == kEntErrCapacity) { if (rc == PFV_OK) rc = PFV_ERR_NOMEM; v = 0; }
        s->ent_sizes[i] = sizes_out[i] = v;
    }
    if (rc == PFV_ERR_FORMAT) return fail(ctx, rc, "coefficient needs more than 15 size bits (src/rle.rs:44)");
    if (rc == PFV_ERR_NOMEM) return fail(ctx, rc, "payload exceeds the capacity given to pfv_enc_entropy_enable");
    return PFV_OK;
}
// Every stream's payload with ONE device-to-host copy: the payloads are gathered back to back on the device (starts
// 16-byte aligned) and land in `out` (ideally page-locked, pfv_host_alloc); offsets_out[s] / sizes_out[s] locate stream
// s in it.  `cap` must hold the sum of the sizes rounded up to 16 each.  Synchronises; errors as pfv_enc_payload_sizes.
PFV_API int pfv_enc_payloads_fetch(pfv_enc_session *s, uint8_t *out, size_t cap, uint32_t *sizes_out, uint64_t *offsets_out)
{
    if (!s || !out || !sizes_out || !offsets_out) return fail(s ? s->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_enc_payloads_fetch: bad argument");
    pfv_ctx *ctx = s->ctx;
    int rc = pfv_enc_payload_sizes(s, sizes_out);
    if (rc) return rc;
    const int S = s->n_streams;
    std::vector<uint32_t> off((size_t)S);
    size_t total = 0;
    for (int i = 0; i < S; i++) {
        off[(size_t)i] = (uint32_t)total;
        offsets_out[i] = total;
        total += ((size_t)sizes_out[i] + 15) & ~(size_t)15;
    }
    if (total > cap || total > 0xfffffff0u) return fail(ctx, PFV_ERR_NOMEM, "pfv_enc_payloads_fetch: output buffer too small");
    if (total == 0) return PFV_OK;
    hipStream_t st = s->ent_stream ? s->ent_stream : ctx->stream;
    if (total > s->ent_packed_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (s->ent_packed) (void)hipFree(s->ent_packed);
        s->ent_packed = nullptr; s->ent_packed_cap = 0;
        const size_t want = total + total / 2;
        HIP_TRY(ctx, hipMalloc((void **)&s->ent_packed, want));
        s->ent_packed_cap = want;
    }
    if (!s->ent_offsets_dev) HIP_TRY(ctx, hipMalloc((void **)&s->ent_offsets_dev, (size_t)S * 4));
    HIP_TRY(ctx, hipMemcpyAsync(s->ent_offsets_dev, off.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
    EntFrame f{};
    f.n_streams = S;
    f.cap_bytes = s->ent_cap;
    f.ones16 = 0x00010001u;
    hipLaunchKernelGGL(k_ent_gather, dim3(32, (unsigned)S), dim3(kEntThreads), 0, st, f, s->ent, s->ent_offsets_dev, s->ent_packed);
    if ((rc = launch_check(ctx, "k_ent_gather"))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, s->ent_packed, total, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));   // also keeps `off` alive long enough
    return PFV_OK;
}

PFV_API const uint8_t *pfv_enc_payload_dev(pfv_enc_session *s, int stream)
{
    if (!s || !s->ent_on || stream < 0 || stream >= s->n_streams) return nullptr;
    return s->ent.payload + (size_t)stream * s->ent_cap;
}
PFV_API size_t pfv_enc_payload_capacity(pfv_enc_session *s) { return s && s->ent_on ? s->ent_cap : 0; }
// Copies the first `nbytes` of one stream's payload to the host (synchronises).
PFV_API int pfv_enc_payload_fetch(pfv_enc_session *s, int stream, uint8_t *out, size_t nbytes)
{
    if (!s || !out) return fail(s ? s->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_enc_payload_fetch: bad argument");
    pfv_ctx *ctx = s->ctx;
    if (!s->ent_on) return fail(ctx, PFV_ERR_STATE, "call pfv_enc_entropy_enable first");
    if (stream < 0 || stream >= s->n_streams || nbytes > s->ent_cap) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_enc_payload_fetch: out of range");
    hipStream_t st = s->ent_stream ? s->ent_stream : ctx->stream;
    if (nbytes) HIP_TRY(ctx, hipMemcpyAsync(out, s->ent.payload + (size_t)stream * s->ent_cap, nbytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return PFV_OK;
}

// ------------------------------------------------------------------ decoder session
PFV_API int pfv_dec_session_create(pfv_ctx *ctx, int width, int height, const int32_t *qtables, int n_qtables,
                                   int n_streams, pfv_dec_session **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_session_create: bad argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (width & 1) || (height & 1) || width > 65535 || height > 65535)
        return fail(ctx, PFV_ERR_BAD_ARG, "width/height must be even (src/frame.rs:13) and fit u16");
    if (!qtables || n_qtables <= 0 || n_qtables > 256 || n_streams <= 0)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_session_create: bad q-table set or stream count");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<QTab> tabs((size_t)n_qtables);
    for (int i = 0; i < n_qtables; i++) {
        int rc = make_qtab(ctx, qtables + (size_t)i * 64, &tabs[i], true);
        if (rc) return rc;
    }
    pfv_dec_session *s = new pfv_dec_session();
    s->ctx = ctx; s->width = width; s->height = height; s->n_streams = n_streams; s->n_qtables = n_qtables;
    s->win_count = n_streams;
    s->lane_mapping = ctx->opt_lane_mapping;
    s->geom = frame_geom(width, height, n_streams);
    size_t pad_bytes = (size_t)s->geom.pad_frame_bytes * n_streams;
    hipError_t e = hipMalloc((void **)&s->qtab_dev, tabs.size() * sizeof(QTab));
    if (e == hipSuccess) e = hipMemcpy(s->qtab_dev, tabs.data(), tabs.size() * sizeof(QTab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fb[0], pad_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fb[1], pad_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->flag_dev, (size_t)n_streams * sizeof(int));
    if (e == hipSuccess) e = hipMemset(s->flag_dev, 0, (size_t)n_streams * sizeof(int));
    if (e != hipSuccess) {
        int rc = hip_fail(ctx, e, "pfv_dec_session_create");
        pfv_dec_session_destroy(s);
        return rc;
    }
    // framebuffer = VideoFrame::new_padded (src/dec.rs:123)
    int rc = init_padded(ctx, s->geom, s->fb[0]);
    if (!rc) rc = init_padded(ctx, s->geom, s->fb[1]);
    if (rc) { pfv_dec_session_destroy(s); return rc; }
    *out = s;
    return PFV_OK;
}

PFV_API void pfv_dec_session_destroy(pfv_dec_session *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    void *bufs[] = {s->qtab_dev, s->fb[0], s->fb[1], s->flag_dev, s->st_coef, s->st_mv, s->st_has, s->st_frames, s->st_idx, s->st_val};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete s;
}

PFV_API int pfv_dec_set_output_dev(pfv_dec_session *s, uint8_t *frames_out_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    s->frames_out = frames_out_dev;
    s->out_stride = 0;
    return PFV_OK;
}
PFV_API int pfv_dec_set_output_strided_dev(pfv_dec_session *s, uint8_t *frames_out_dev, size_t stride_bytes)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (stride_bytes && stride_bytes < (size_t)s->geom.src_frame_bytes)
        return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_set_output_strided_dev: stride below pfv_frame_bytes (the slots' frames would overlap)");
    s->frames_out = frames_out_dev;
    s->out_stride = stride_bytes;
    return PFV_OK;
}
PFV_API int pfv_dec_session_set_window(pfv_dec_session *s, int first, int count)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (first < 0 || count <= 0 || first > s->n_streams - count) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_session_set_window: window outside [0, n_streams)");
    s->win_first = first; s->win_count = count;
    return PFV_OK;
}
static bool dec_full_window(const pfv_dec_session *s) { return s->win_first == 0 && s->win_count == s->n_streams && s->out_stride == 0; }

// geometry of the retframe output: stride between the slots' frames; 16-byte vector stores need an aligned base and stride
static FrameGeom dec_out_geom(const pfv_dec_session *s, FrameGeom g, const uint8_t *out_base)
{
    if (s->out_stride) {
        g.src_frame_bytes = (long)s->out_stride;
        if (s->out_stride % 16)
            for (int i = 0; i < 3; i++) g.p[i].fast_src = 0;
    }
    return with_base_alignment(g, out_base);
}
// the decode kernels can write the retframe themselves only with 16-byte vector stores
static bool fused_output_ok(const pfv_dec_session *s)
{
    if (!s->frames_out) return false;
    FrameGeom g = dec_out_geom(s, s->geom, s->frames_out);
    return g.p[0].fast_src && g.p[1].fast_src && g.p[2].fast_src;
}

static int dec_geom(pfv_dec_session *s, const uint8_t qidx[3], FrameGeom *g)
{
    if (!qidx) return fail(s->ctx, PFV_ERR_BAD_ARG, "qidx is null");
    *g = s->geom;
    for (int i = 0; i < 3; i++) {
        if (qidx[i] >= s->n_qtables)
            return fail(s->ctx, PFV_ERR_FORMAT, "q-table index out of range (the reference panics: src/dec.rs:249-251)");
        g->p[i].qsel = qidx[i];
    }
    return PFV_OK;
}
static int dec_crop_win(pfv_dec_session *s, int first, int count, uint8_t *frames_out_dev, size_t out_stride);

// Slots [first, first + count) of a session (see enc_launch): same launch on shifted bases, ping-pong index untouched.
static int dec_launch(pfv_dec_session *s, bool pframe, int first, int count, const int8_t *mv_dev, const uint8_t *has_dev, const DecCoefs &coefs,
                      const uint8_t qidx[3])
{
    pfv_ctx *ctx = s->ctx;
    FrameGeom g;
    int rc = dec_geom(s, qidx, &g);
    if (rc) return rc;
    const size_t ostride = s->out_stride ? s->out_stride : (size_t)s->geom.src_frame_bytes;
    const size_t mb0 = (size_t)first * (size_t)s->geom.mbs_per_frame, pad0 = (size_t)first * (size_t)s->geom.pad_frame_bytes;
    const bool fused = fused_output_ok(s);
    uint8_t *crop = fused ? s->frames_out + (size_t)first * ostride : nullptr;
    g = dec_out_geom(s, g, s->frames_out);
    g.n_streams = count;
    const int nxt = s->cur ^ 1;
    if (pframe) {
        launch_dec_pframe(ctx, use_small_grid(s->lane_mapping, g), g, mv_dev + mb0 * 2, has_dev + mb0, coefs.shifted((size_t)first, (size_t)s->geom.mbs_per_frame),
                          s->fb[s->cur] + pad0, s->fb[nxt] + pad0, s->qtab_dev, s->flag_dev + first, crop);
        rc = launch_check(ctx, "k_dec_pframe");
    } else {
        launch_dec_iframe(ctx, use_small_grid(s->lane_mapping, g), g, coefs.shifted((size_t)first, (size_t)s->geom.mbs_per_frame), s->fb[nxt] + pad0, s->qtab_dev, crop);
        rc = launch_check(ctx, "k_dec_iframe");
    }
    return rc;
}

}  // extern "C"
// one frame operation on the session's window: the launch, the ping-pong, the separate crop pass where the fused one does not apply
static int dec_step(pfv_dec_session *s, bool pframe, const int8_t *mv_dev, const uint8_t *has_coef_dev, const DecCoefs &coefs, const uint8_t qidx[3])
{
    pfv_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dec_launch(s, pframe, s->win_first, s->win_count, mv_dev, has_coef_dev, coefs, qidx);
    if (rc) return rc;
    s->cur ^= 1;
    if (s->frames_out && !fused_output_ok(s)) return dec_crop_win(s, s->win_first, s->win_count, s->frames_out, s->out_stride);
    return PFV_OK;
}
extern "C" {

PFV_API int pfv_dec_iframe_dev(pfv_dec_session *s, const int16_t *coef_dev, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!coef_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_iframe_dev: null buffer");
    return dec_step(s, false, nullptr, nullptr, coef_dev, qidx);
}

PFV_API int pfv_dec_pframe_dev(pfv_dec_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev,
                               const int16_t *coef_dev, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!mv_dev || !has_coef_dev || !coef_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_pframe_dev: null buffer");
    return dec_step(s, true, mv_dev, has_coef_dev, coef_dev, qidx);
}

// The same two operations on COEFFICIENT LISTS (round 5; the form the stream decoders' entropy stage produces, see pfv_hip.h): per slot of
// the session's window a pointer to its list of entries and, per macroblock (+ 1), the number of entries before it.  Same result as the
// dense call on the expanded arrays.  The counts must be what pfv_coef_lists_from_dense / the decoders produce (ascending, within the
// slot's list): they are the kernels' loop bounds and are not validated on the device.
PFV_API int pfv_dec_iframe_lists_dev(pfv_dec_session *s, const uint32_t *const *entries_dev, const uint32_t *counts_dev, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!entries_dev || !counts_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_iframe_lists_dev: null buffer");
    return dec_step(s, false, nullptr, nullptr, DecCoefs(entries_dev, counts_dev), qidx);
}
PFV_API int pfv_dec_pframe_lists_dev(pfv_dec_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev, const uint32_t *const *entries_dev,
                                     const uint32_t *counts_dev, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    if (!mv_dev || !has_coef_dev || !entries_dev || !counts_dev) return fail(s->ctx, PFV_ERR_BAD_ARG, "pfv_dec_pframe_lists_dev: null buffer");
    return dec_step(s, true, mv_dev, has_coef_dev, DecCoefs(entries_dev, counts_dev), qidx);
}
// Host helper: one frame's dense coefficients ([total_blocks][256]) as a coefficient list.  has_coef (nullable: every macroblock) says which
// macroblocks are read.  entries_out has room for `cap` entries, counts_out for total_blocks + 1 counts; *n_out = entries written.
// Returns 1 when more than `cap` entries would be needed (total_blocks x 256 always suffices).
PFV_API int pfv_coef_lists_from_dense(const int16_t *coef, const uint8_t *has_coef, int total_blocks, uint32_t *entries_out, size_t cap, uint32_t *counts_out,
                                      size_t *n_out)
{
    if (!coef || !entries_out || !counts_out || !n_out || total_blocks <= 0) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_coef_lists_from_dense: bad argument");
    ListSink sink{entries_out, cap, counts_out, (size_t)total_blocks};
    bool full = false;
    for (size_t b = 0; b < (size_t)total_blocks && !full; b++) {
        if (has_coef && !has_coef[b]) continue;
        for (size_t i = 0; i < 256 && !full; i++)
            if (coef[b * 256 + i]) full = !sink.put(b * 256 + i, coef[b * 256 + i]);
    }
    sink.finish();
    *n_out = sink.n;
    return full ? 1 : PFV_OK;
}

// one flag per slot (k_dec_pframe raises flag[stream]); PFV_ERR_BAD_MV when any is set, all cleared
static int dec_check_flags(pfv_dec_session *s, std::vector<int> *which)
{
    pfv_ctx *ctx = s->ctx;
    s->flags_host.assign((size_t)s->n_streams, 0);
    HIP_TRY(ctx, hipMemcpyAsync(s->flags_host.data(), s->flag_dev, (size_t)s->n_streams * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    bool any = false;
    for (int k = 0; k < s->n_streams; k++) any = any || s->flags_host[(size_t)k] != 0;
    if (which) *which = s->flags_host;
    if (any) {
        HIP_TRY(ctx, hipMemsetAsync(s->flag_dev, 0, (size_t)s->n_streams * sizeof(int), ctx->stream));
        return fail(ctx, PFV_ERR_BAD_MV, "motion vector points outside the reference plane (src/common.rs:258-259)");
    }
    return PFV_OK;
}
PFV_API int pfv_dec_check(pfv_dec_session *s)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    return dec_check_flags(s, nullptr);
}

static int dec_staging(pfv_dec_session *s)
{
    pfv_ctx *ctx = s->ctx;
    if (s->st_coef) return PFV_OK;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMalloc((void **)&s->st_coef, n * 512));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_mv, n * 2));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_has, n));
    HIP_TRY(ctx, hipMalloc((void **)&s->st_frames, (size_t)s->geom.src_frame_bytes * s->n_streams));
    return PFV_OK;
}

PFV_API int pfv_dec_iframe(pfv_dec_session *s, const int16_t *coef, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!coef) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_iframe: null buffer");
    if (!dec_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_dec_iframe: the host-buffer entry points work on all slots, packed (reset the window / output stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dec_staging(s);
    if (rc) return rc;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMemcpyAsync(s->st_coef, coef, n * 512, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pfv_dec_iframe_dev(s, s->st_coef, qidx))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_dec_pframe(pfv_dec_session *s, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef,
                           const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!mv || !has_coef || !coef) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_pframe: null buffer");
    if (!dec_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_dec_pframe: the host-buffer entry points work on all slots, packed (reset the window / output stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dec_staging(s);
    if (rc) return rc;
    size_t n = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMemcpyAsync(s->st_coef, coef, n * 512, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(s->st_mv, mv, n * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(s->st_has, has_coef, n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pfv_dec_pframe_dev(s, s->st_mv, s->st_has, s->st_coef, qidx))) return rc;
    return pfv_dec_check(s);
}

// Sparse forms of pfv_dec_iframe / pfv_dec_pframe: the non-zero coefficients as (flat index into
// [stream][macroblock][256], value) pairs, everything else zero.  Same result as the dense call on the expanded array.
static int dec_upload_sparse(pfv_dec_session *s, const uint32_t *idx, const int16_t *val, size_t n)
{
    pfv_ctx *ctx = s->ctx;
    int rc = dec_staging(s);
    if (rc) return rc;
    const size_t total = (size_t)s->geom.mbs_per_frame * s->n_streams * 256;
    if (n > total || total > 0xffffffffull) return fail(ctx, PFV_ERR_BAD_ARG, "sparse coefficient list longer than the frame");
    if (n > s->st_sparse_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (s->st_idx) (void)hipFree(s->st_idx);
        if (s->st_val) (void)hipFree(s->st_val);
        s->st_idx = nullptr; s->st_val = nullptr; s->st_sparse_cap = 0;
        const size_t cap = std::max(n, total / 8);
        HIP_TRY(ctx, hipMalloc((void **)&s->st_idx, cap * 4));
        HIP_TRY(ctx, hipMalloc((void **)&s->st_val, cap * 2));
        s->st_sparse_cap = cap;
    }
    HIP_TRY(ctx, hipMemsetAsync(s->st_coef, 0, total * 2, ctx->stream));
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(s->st_idx, idx, n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(s->st_val, val, n * 2, hipMemcpyHostToDevice, ctx->stream));
        const unsigned blocks = (unsigned)std::min<size_t>((n + kThreads - 1) / kThreads, 4096);
        hipLaunchKernelGGL(k_scatter_coef, dim3(blocks), dim3(kThreads), 0, ctx->stream, s->st_idx, s->st_val, (uint32_t)n, (uint32_t)total,
                           s->st_coef);
        if ((rc = launch_check(ctx, "k_scatter_coef"))) return rc;
    }
    return PFV_OK;
}
PFV_API int pfv_dec_iframe_sparse(pfv_dec_session *s, const uint32_t *idx, const int16_t *val, size_t n, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (n && (!idx || !val)) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_iframe_sparse: null buffer");
    if (!dec_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_dec_iframe_sparse: the host-buffer entry points work on all slots, packed (reset the window / output stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dec_upload_sparse(s, idx, val, n);
    if (rc) return rc;
    if ((rc = pfv_dec_iframe_dev(s, s->st_coef, qidx))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}
PFV_API int pfv_dec_pframe_sparse(pfv_dec_session *s, const int8_t *mv, const uint8_t *has_coef, const uint32_t *idx,
                                  const int16_t *val, size_t n, const uint8_t qidx[3])
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!mv || !has_coef || (n && (!idx || !val))) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_pframe_sparse: null buffer");
    if (!dec_full_window(s)) return fail(ctx, PFV_ERR_STATE, "pfv_dec_pframe_sparse: the host-buffer entry points work on all slots, packed (reset the window / output stride)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = dec_upload_sparse(s, idx, val, n);
    if (rc) return rc;
    const size_t nmb = (size_t)s->geom.mbs_per_frame * s->n_streams;
    HIP_TRY(ctx, hipMemcpyAsync(s->st_mv, mv, nmb * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(s->st_has, has_coef, nmb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pfv_dec_pframe_dev(s, s->st_mv, s->st_has, s->st_coef, qidx))) return rc;
    return pfv_dec_check(s);
}

static int dec_crop_win(pfv_dec_session *s, int first, int count, uint8_t *frames_out_dev, size_t out_stride)
{
    pfv_ctx *ctx = s->ctx;
    FrameGeom g = s->geom;
    g.n_streams = count;
    if (out_stride) {
        g.src_frame_bytes = (long)out_stride;
        if (out_stride % 16)
            for (int i = 0; i < 3; i++) g.p[i].fast_src = 0;
    }
    uint8_t *dst = frames_out_dev + (size_t)first * (size_t)g.src_frame_bytes;
    g = with_base_alignment(g, dst);
    dim3 grid(128, 3, g.n_streams);
    hipLaunchKernelGGL(k_crop_frames, grid, dim3(kThreads), 0, ctx->stream, g, s->fb[s->cur] + (size_t)first * (size_t)g.pad_frame_bytes, dst);
    return launch_check(ctx, "k_crop_frames");
}
PFV_API int pfv_dec_get_frame_dev(pfv_dec_session *s, uint8_t *frames_out_dev)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames_out_dev) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_get_frame_dev: null buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return dec_crop_win(s, 0, s->n_streams, frames_out_dev, 0);
}

PFV_API int pfv_dec_get_frame(pfv_dec_session *s, uint8_t *frames_out)
{
    if (!s) return fail(nullptr, PFV_ERR_BAD_ARG, "null session");
    pfv_ctx *ctx = s->ctx;
    if (!frames_out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_dec_get_frame: null buffer");
    int rc = dec_staging(s);
    if (rc) return rc;
    if ((rc = pfv_dec_get_frame_dev(s, s->st_frames))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(frames_out, s->st_frames, (size_t)s->geom.src_frame_bytes * s->n_streams,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

PFV_API int pfv_dec_framebuffer(pfv_dec_session *s, uint8_t *out_host)
{
    if (!s || !out_host) return fail(s ? s->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_dec_framebuffer: bad argument");
    pfv_ctx *ctx = s->ctx;
    HIP_TRY(ctx, hipMemcpyAsync(out_host, s->fb[s->cur], (size_t)s->geom.pad_frame_bytes * s->n_streams,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFV_OK;
}

}  // extern "C"

// ================================================================== stream-level session objects
// enc::Encoder<W> (src/enc.rs:12-188) with W = an in-memory byte vector (the reference's tests use
// Cursor<Vec<u8>>, src/lib.rs:319-321), dec::Decoder<R> (src/dec.rs:15-224) with R = a caller-owned byte slice.
// Page-locked host staging (hipHostMalloc): PCIe copies from / to these run at link rate without the runtime's
// bounce through its own pinned chunks; where page-locking is refused the buffer is ordinary memory.
template <class T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    bool pinned = false;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release()
    {
        if (p && pinned) (void)hipHostFree(p);
        else if (p) free(p);
        p = nullptr; n = 0;
    }
    bool resize(size_t count)
    {
        if (count <= n) return true;
        release();
        if (hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault) == hipSuccess) {
            pinned = true;
        } else {   // locked-memory limits: pageable memory still works, the copies just bounce through the runtime
            (void)hipGetLastError();
            p = (T *)malloc(count * sizeof(T));
            pinned = false;
            if (!p) return false;
        }
        n = count;
        return true;
    }
    T *data() { return p; }
    size_t size() const { return n; }
    void swap(PinnedBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(pinned, o.pinned); }
};

struct pfv_encoder {
    pfv_ctx *ctx = nullptr;
    pfv_enc_session *hot = nullptr;
    int width = 0, height = 0, framerate = 0, total_blocks = 0;
    bool finished = false;
    bool device_entropy = true;            // payloads built by the k_ent_* kernels instead of serialize_*frame on the host
    const uint8_t *plane[3] = {nullptr, nullptr, nullptr};   // device path: the caller's planes of the frame being encoded
    bool poisoned = false;                 // a frame failed after prev_frame had moved on: the next frame must be an i-frame
    std::vector<uint8_t> out;              // the writer: bytes produced and not yet handed over (pfv_encoder_drain)
    std::vector<uint8_t> drained;          // what the last pfv_encoder_drain handed over
    PinnedBuf<uint8_t> frame;              // packed Y|U|V staging
    PinnedBuf<int16_t> coef;               // host entropy path only
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    PinnedBuf<uint8_t> payload;            // device entropy path: packet payload landing zone
};

// One step of Decoder::advance_frame's packet loop (src/dec.rs:169-224), found by the header scanner.  FRAME events are
// parsed (bits -> coefficients / block headers, dec.rs:226-296, 328-417) ahead of their turn by worker threads: packets
// are independent bit streams, only the device decode behind them is sequential.
// ------------------------------------------------------------------ the decoders' entropy stage on the device: host half
// What the host reads of a packet for the k_entd_* kernels (pfv_entdec_kernels.hip): its first 19 bytes -- the table (-> the tree's codes) and
// the q indices.  A p-frame's block headers are read on the device since round 5 (k_hdr_*: motion vectors, has_coeff, the first bit of the run
// streams), the list of coded macroblocks is made there from the has_coeff bytes (k_entd_coded).
// The payload is copied to `bytes_dst` (page-locked staging, >= plen + 16 bytes).  The caller has set k.byte_off / k.frame_off.
// header workgroups (k_hdr_*) of a p-frame packet: 2 048 bits each, as many as its headers can take (16 bits per macroblock) or its payload has
static inline uint32_t entd_hdr_wgs(size_t tb, size_t plen)
{
    const size_t bits = plen * 8 > kHdrBit0 ? plen * 8 - kHdrBit0 : 0;
    return (uint32_t)((std::min(bits, tb * 16) + kHdrWgBits - 1) / kHdrWgBits);
}
struct EntdPrep {
    int rc = 0;                  // a status the host parser would have returned before it read any run (header, q index, truncated block headers)
    bool host_parse = false;     // the host parser has to read this packet (degenerate code table, 512 MiB or more, no bits behind the headers)
    uint8_t qidx[3] = {0, 0, 0};
};
static EntdPrep entd_prepare(const uint8_t *payload, uint32_t plen, int type, size_t tb, int n_qtables, uint32_t sub_bits, EdPacket &k, uint8_t *bytes_dst)
{
    EntdPrep p;
    k.total_bits = k.bit0 = k.total_coefs = k.n_sub = k.sub_first = k.grp_first = k.list_cap = 0;
    k.org = k.first_sub = k.hdr_first = k.hdr_wgs = 0;
    k.sub_bits = sub_bits;
    k.pframe = type == 2 ? 1u : 0u;
    k.total_blocks = (uint32_t)tb;
    memset(k.code_val, 0, sizeof k.code_val);
    memset(k.code_len, 0, sizeof k.code_len);
    BitSource r(payload, plen);
    PacketHead h;
    p.rc = parse_head(r, h, n_qtables);
    if (p.rc) return p;
    memcpy(p.qidx, h.qidx, 3);
    int n_syms = 0;
    for (uint8_t t : h.table) n_syms += t != 0;
    const uint64_t bits = (uint64_t)plen * 8, bit0 = r.position();      // behind the table and the q indices: bit 152
    // zero-length codes / no bits left / 64 MiB and more: a run costs two bits or more and covers at most 16 coefficients, so below 2^29 bits
    // the kernels' counters (coefficients and values per packet, 32 bits each, summed side by side in one 64-bit word) cannot overflow
    if (n_syms < 2 || bits >= (1ull << 29) || bit0 >= bits) { p.host_parse = true; return p; }
    HuffmanTree tree(h.table);
    for (int s = 0; s < 16; s++) {
        k.code_val[s] = (uint16_t)tree.code((uint8_t)s).val;
        k.code_len[s] = (uint8_t)tree.code((uint8_t)s).len;
    }
    k.total_bits = (uint32_t)bits;
    k.bit0 = k.org = (uint32_t)bit0;
    k.n_sub = (uint32_t)((bits - bit0 + sub_bits - 1) / sub_bits);
    if (type == 2) {
        // the block headers (src/dec.rs:351-372) are read on the device (k_hdr_*): where the run streams start, how many macroblocks are coded
        // and what the list can need is written into the descriptor there; the subsequences are counted from bit 152
        k.hdr_wgs = entd_hdr_wgs(tb, plen);
    } else {
        k.total_coefs = (uint32_t)(tb * 256);
        k.list_cap = (uint32_t)std::min<uint64_t>(tb * 256, (bits - bit0) / 3 + 1);   // <= entd_pool_cap(tb, plen): the room the caller set aside
    }
    memcpy(bytes_dst, payload, plen);
    memset(bytes_dst + plen, 0, 16);
    return p;
}
// Entries a packet's coefficient list can need, known before any of it is read: a value costs three bits or more (two tree codes of a bit or
// more -- tables of fewer than two symbols go to the host parser -- and coeff_size >= 1 value bits), and there are no more values than
// coefficients.  Rounded up to whole 16-byte lines so that the lists of a pool start aligned.
static inline size_t entd_pool_cap(size_t tb, size_t plen) { return (std::min(tb * 256, plen * 8 / 3 + 1) + 3) & ~(size_t)3; }

// Device side of the coefficient lists of `frames` frames (pfv_device.h: CoefLists): a pool of entries the frames' lists are cut from, the
// table of list pointers the decode kernels index by slot, the frames' counts.  A list that does not fit its place in the pool -- only a
// packet the HOST parser read can need more than entd_pool_cap (a one-symbol table: values of one or two bits) -- gets a buffer of its own
// for the life of the batch (spill).
struct ListPool {
    uint32_t *ent = nullptr; size_t ent_cap = 0;       // entries
    uint32_t **ptr_dev = nullptr;                      // [frames]
    uint32_t *counts_dev = nullptr;                    // [frames][tb + 1]
    size_t frames = 0, tb = 0;
    PinnedBuf<uint32_t *> ptr_host;
    std::vector<uint32_t *> spill;
    long spilled = 0;                                  // lists that got a buffer of their own so far
    int create(pfv_ctx *ctx, size_t n_frames, size_t total_blocks, size_t entries)
    {
        frames = n_frames; tb = total_blocks;
        HIP_TRY(ctx, hipMalloc((void **)&ptr_dev, n_frames * sizeof(uint32_t *)));
        HIP_TRY(ctx, hipMalloc((void **)&counts_dev, n_frames * (total_blocks + 1) * sizeof(uint32_t)));
        if (entries) { HIP_TRY(ctx, hipMalloc((void **)&ent, entries * sizeof(uint32_t))); ent_cap = entries; }
        if (!ptr_host.resize(n_frames)) return fail(ctx, PFV_ERR_NOMEM, "pinned list-pointer staging");
        for (size_t f = 0; f < n_frames; f++) ptr_host.data()[f] = nullptr;
        return PFV_OK;
    }
    // room for `entries` in the pool; the caller has made sure nothing on the device still uses it
    int room(pfv_ctx *ctx, size_t entries)
    {
        if (entries <= ent_cap) return PFV_OK;
        if (ent) { (void)hipFree(ent); ent = nullptr; ent_cap = 0; }
        entries += entries / 4;
        HIP_TRY(ctx, hipMalloc((void **)&ent, entries * sizeof(uint32_t)));
        ent_cap = entries;
        return PFV_OK;
    }
    void drop_spill()
    {
        for (uint32_t *p : spill) (void)hipFree(p);
        spill.clear();
    }
    void destroy()
    {
        drop_spill();
        for (void *p : {(void *)ent, (void *)ptr_dev, (void *)counts_dev})
            if (p) (void)hipFree(p);
        ent = nullptr; ptr_dev = nullptr; counts_dev = nullptr; ent_cap = 0;
    }
    DecCoefs coefs(size_t first_frame = 0) const { return DecCoefs(ptr_dev + first_frame, counts_dev + first_frame * (tb + 1)); }
};

// A packet through the HOST parser into list form, for a decoder whose coefficients travel as lists: entries and counts into page-locked
// staging (`ent` with room for `cap` entries, `counts` [tb + 1]).  kSinkFull: more than `cap` entries (parse again with room for tb x 256).
static int parse_to_lists(const uint8_t *payload, size_t plen, int type, size_t tb, int n_qtables, int8_t *mv, uint8_t *has, uint32_t *ent, size_t cap, uint32_t *counts,
                          size_t *n_out, uint8_t qidx[3])
{
    ListSink sink{ent, cap, counts, tb};
    const int rc = type == 2 ? parse_pframe_to(payload, plen, (int)tb, n_qtables, mv, has, sink, qidx) : parse_iframe_to(payload, plen, (int)tb, n_qtables, sink, qidx);
    sink.finish();
    *n_out = sink.n;
    return rc;
}
// ... and onto the device, in frame `f`'s place of the pool (or a buffer of its own when it is longer than the place: `place_cap` entries),
// on `stream`; the staging is free again when the stream has passed this point
static int upload_lists(pfv_ctx *ctx, ListPool &lp, size_t f, size_t place_cap, const uint32_t *ent, size_t n, const uint32_t *counts, hipStream_t stream)
{
    uint32_t *dst = lp.ptr_host.data()[f];
    if (n > place_cap || !dst) {
        HIP_TRY(ctx, hipMalloc((void **)&dst, std::max<size_t>(n, 1) * sizeof(uint32_t)));
        lp.spill.push_back(dst);
        lp.spilled++;
        lp.ptr_host.data()[f] = dst;
        HIP_TRY(ctx, hipMemcpyAsync(lp.ptr_dev + f, lp.ptr_host.data() + f, sizeof(uint32_t *), hipMemcpyHostToDevice, stream));
    }
    if (n) HIP_TRY(ctx, hipMemcpyAsync(dst, ent, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(lp.counts_dev + f * (lp.tb + 1), counts, (lp.tb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    return PFV_OK;
}

// the launches of one window: np packets from b.packet0 on, ng workgroups from b.group0 on (b.groups already points at the first of them)
static void entd_launch(hipStream_t stream, const EdBufs &b, unsigned np, unsigned ng, unsigned max_hdr_wgs, int launches, int inner)
{
    if (max_hdr_wgs) {   // the p-frames' block headers first: they complete the packet descriptors the kernels below read
        hipLaunchKernelGGL(k_hdr_map, dim3(max_hdr_wgs, np), dim3(kEdThreads), 0, stream, b);
        hipLaunchKernelGGL(k_hdr_scan, dim3(np), dim3(kEdThreads), 0, stream, b);
        hipLaunchKernelGGL(k_hdr_emit, dim3(max_hdr_wgs, np), dim3(kEdThreads), 0, stream, b);
    }
    hipLaunchKernelGGL(k_entd_coded, dim3(np), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_sync, dim3(ng), dim3(kEdThreads), 0, stream, b, inner);                      // every subsequence, settled inside the workgroups
    for (int round = 1; round < launches; round++)                                                           // the seams between them (a second pass finds nothing, as a rule)
        hipLaunchKernelGGL(k_entd_fix, dim3((ng + kEdFixThreads - 1) / kEdFixThreads), dim3(kEdFixThreads), 0, stream, b, (uint32_t)ng);
    hipLaunchKernelGGL(k_entd_verify, dim3(ng), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_prefix, dim3(np), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_emit, dim3(ng), dim3(kEdThreads), 0, stream, b);
}

struct DecEvent {
    enum Kind { FRAME, DROP, END, ERROR } kind = END;
    enum State { FREE, QUEUED, RUNNING, DONE } state = FREE;
    int rc = 0;                          // ERROR: the status to return; FRAME: parse result
    const char *msg = "";
    uint8_t type = 0;                    // FRAME: 1 = i-frame, 2 = p-frame
    size_t pos_after = 0;                // stream position once this event has been consumed
    const uint8_t *payload = nullptr;
    uint32_t plen = 0;
    uint8_t qidx[3] = {0, 0, 0};
    PinnedBuf<int16_t> coef;             // dense form: only when the sparse list overflowed
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    PinnedBuf<uint32_t> idx;             // sparse form: non-zero coefficients as (flat index, value)
    PinnedBuf<int16_t> val;
    size_t n_sparse = 0;
    bool dense = false;
    // device-entropy form (PFV_OPT_ENTROPY_DECODE): what entd_prepare left for the k_entd_* kernels instead of a parsed packet
    bool dev_form = false, host_parse = false;
    PinnedBuf<uint8_t> bytes;            // the payload (+ 16)
    PinnedBuf<EdPacket> pk;              // 1
    PinnedBuf<uint2> groups;             // workgroups of the packet
};

// switches, shape and counters of the device entropy stage in pfv_decoder / pfv_batch_decoder (the buffers: DecWindow)
struct DecEntd {
    std::atomic<bool> on{false};         // read by the parser threads; cleared by the caller's thread when the window sets cannot be made (AUTO: the host parser takes over)
    bool force = false;                  // force: every packet (PFV_ENTROPY_DECODE_DEVICE); otherwise payloads of kDecEntdMinBytes and more
    bool ready = false;                  // the window stream and the window sets exist: made by the first packet / step that takes the device form
    //                                      (a decoder of small packets never needs them), entd_windows_make
    uint32_t sub_bits = kEdSubBits;
    int launches = 3, inner = kEdInner;
    long packets_dev = 0, packets_host = 0;
};
constexpr uint32_t kDecEntdMinBytes = 64 * 1024;   // below this the launches cost more than the host parser needs for the packet

// device side of one packet's window in pfv_decoder.  Two alternate: the window of the NEXT packet (uploads, k_entd_*, status) runs on a
// second stream under the decode launch and the frame download of the current one.
struct DecWindow {
    uint8_t *bytes_dev = nullptr; size_t bytes_cap = 0;
    uint2 *groups_dev = nullptr; size_t groups_cap = 0;
    uint32_t *sub_dev = nullptr; size_t sub_cap = 0;
    EdPacket *pk_dev = nullptr;
    uint32_t *status_dev = nullptr, *coded_dev = nullptr;
    unsigned long long *wgsum_dev = nullptr; size_t wgsum_cap = 0;
    uint32_t *hdr_maps_dev = nullptr; size_t hdr_maps_cap = 0;      // k_hdr_*: [header workgroup][8]
    uint4 *hdr_start_dev = nullptr; size_t hdr_start_cap = 0;       // [header workgroup]
    ListPool lists;                      // the window's coefficients: one list per packet (pfv_device.h: CoefLists)
    std::vector<size_t> list_room;       // per packet: the size of its list's place in the pool
    int8_t *mv_dev = nullptr;
    uint8_t *has_dev = nullptr;
    PinnedBuf<uint32_t> status_host;
    hipEvent_t done = nullptr;
    DecEvent *owner = nullptr;           // the packet whose window is enqueued / was decoded from this set
    void destroy()
    {
        for (void *p : {(void *)bytes_dev, (void *)pk_dev, (void *)status_dev, (void *)coded_dev, (void *)groups_dev, (void *)sub_dev, (void *)wgsum_dev, (void *)mv_dev, (void *)has_dev,
                        (void *)hdr_maps_dev, (void *)hdr_start_dev})
            if (p) (void)hipFree(p);
        lists.destroy();
        if (done) (void)hipEventDestroy(done);
    }
};

// The window stream and the fixed-size part of every window set, for S packets per window: on the caller's thread, when the first packet (step)
// takes the device form.
template <size_t N>
static int entd_windows_make(pfv_ctx *ctx, DecEntd &v, DecWindow (&win)[N], hipStream_t *stream, size_t S, size_t tb)
{
    if (v.ready) return PFV_OK;
    hipError_t he = *stream ? hipSuccess : hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    bool host_ok = true;
    for (DecWindow &w : win) {
        if (he == hipSuccess && !w.pk_dev) he = hipMalloc((void **)&w.pk_dev, S * sizeof(EdPacket));
        if (he == hipSuccess && !w.status_dev) he = hipMalloc((void **)&w.status_dev, S * sizeof(uint32_t));
        if (he == hipSuccess && !w.coded_dev) he = hipMalloc((void **)&w.coded_dev, S * tb * sizeof(uint32_t));
        if (he == hipSuccess && !w.lists.ptr_dev && w.lists.create(ctx, S, tb, 0) != PFV_OK) he = hipErrorOutOfMemory;
        if (he == hipSuccess && !w.mv_dev) he = hipMalloc((void **)&w.mv_dev, S * tb * 2);
        if (he == hipSuccess && !w.has_dev) he = hipMalloc((void **)&w.has_dev, S * tb);
        if (he == hipSuccess && !w.done) he = hipEventCreateWithFlags(&w.done, hipEventDisableTiming);
        host_ok = host_ok && w.status_host.resize(S);
    }
    if (he != hipSuccess) return hip_fail(ctx, he, "device entropy stage: window sets");
    if (!host_ok) return fail(ctx, PFV_ERR_NOMEM, "device entropy stage: pinned status words");
    v.ready = true;
    return PFV_OK;
}
// host staging of one packet the host parser reads into list form (a decoder whose coefficients travel as lists)
struct ListStage {
    PinnedBuf<uint32_t> ent, counts;
    size_t n = 0;
    // kSinkFull cannot come back: a list of the place's size is tried first, then one with room for every coefficient
    int parse(const uint8_t *payload, size_t plen, int type, size_t tb, int n_qtables, int8_t *mv, uint8_t *has, size_t place_cap, uint8_t qidx[3])
    {
        if (!ent.resize(std::max<size_t>(place_cap, 4)) || !counts.resize(tb + 1)) return PFV_ERR_NOMEM;
        int rc = parse_to_lists(payload, plen, type, tb, n_qtables, mv, has, ent.data(), place_cap, counts.data(), &n, qidx);
        if (rc != kSinkFull) return rc;
        if (!ent.resize(tb * 256)) return PFV_ERR_NOMEM;
        return parse_to_lists(payload, plen, type, tb, n_qtables, mv, has, ent.data(), tb * 256, counts.data(), &n, qidx);
    }
};

constexpr int kDecWindows = 4;           // pfv_decoder: windows in flight -- the packet being decoded and up to three behind it
struct pfv_decoder {
    DecEntd entd;                        // switches, shape and counters of the device entropy stage (its buffers: win[])
    DecWindow win[kDecWindows];
    ListStage hp;                        // a packet the device stage left to the host parser
    hipStream_t win_stream = nullptr;
    pfv_ctx *ctx = nullptr;
    pfv_dec_session *hot = nullptr;
    const uint8_t *data = nullptr;
    size_t len = 0, pos = 0, reset_pos = 0;
    int width = 0, height = 0, framerate = 0, n_qtables = 0, total_blocks = 0;
    bool eof = false;
    double delta_accum = 0.0;
    PinnedBuf<uint8_t> retframe;           // Y|U|V, unpadded (src/dec.rs:22)
    uint8_t *frame_dev = nullptr;          // pfv_decoder_set_output_device: the retframe in device memory instead
    // look-ahead: ring of events in stream order, [head, head + count)
    std::vector<std::unique_ptr<DecEvent>> ring;
    size_t head = 0, count = 0;
    size_t scan_pos = 0;
    bool scan_stop = false;                // an END / ERROR event is pending: nothing is scanned past it
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    bool quit = false;
};

static void put_u16(std::vector<uint8_t> &o, unsigned v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
static void put_u32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_packet(std::vector<uint8_t> &o, uint8_t type, const std::vector<uint8_t> *payload)
{
    o.push_back(type);
    put_u32(o, payload ? (uint32_t)payload->size() : 0u);
    if (payload) o.insert(o.end(), payload->begin(), payload->end());
}

extern "C" {

// Encoder::new (src/enc.rs:37-73): q-tables from quality, prev_frame = new_padded, write_header (:190-219)
PFV_API int pfv_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, pfv_encoder **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_encoder_create: bad argument");
    *out = nullptr;
    if (framerate < 0 || framerate > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "framerate must fit u16 (src/enc.rs:197)");
    pfv_enc_session *hot = nullptr;
    int rc = pfv_enc_session_create(ctx, width, height, quality, 1, &hot);
    if (rc) return rc;
    pfv_encoder *e = new pfv_encoder();
    e->ctx = ctx; e->hot = hot; e->width = width; e->height = height; e->framerate = framerate;
    e->total_blocks = pfv_total_blocks(width, height);
    if (!e->frame.resize(pfv_frame_bytes(width, height))) {
        pfv_encoder_destroy(e);
        return fail(ctx, PFV_ERR_NOMEM, "pfv_encoder_create: pinned staging");
    }
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], nullptr);
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};      // common.rs:1
    e->out.insert(e->out.end(), magic, magic + 8);
    put_u32(e->out, 211);                                                      // common.rs:2
    put_u16(e->out, (unsigned)width); put_u16(e->out, (unsigned)height); put_u16(e->out, (unsigned)framerate);
    put_u16(e->out, 4);
    for (int t = 0; t < 4; t++)                                                // intra_l, intra_c, inter_l, inter_c
        for (int i = 0; i < 64; i++) put_u16(e->out, (unsigned)q[t][i]);
    *out = e;
    return PFV_OK;
}

static int pack_frame(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!y || !u || !v) return fail(e->ctx, PFV_ERR_BAD_ARG, "null plane");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:80)");
    if (e->device_entropy) {   // the planes go up from where they lie (encode_on_device): no packing copy -- it was half of a 4K frame's time
        e->plane[0] = y; e->plane[1] = u; e->plane[2] = v;
        return PFV_OK;
    }
    size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
    memcpy(e->frame.data(), y, ny);
    memcpy(e->frame.data() + ny, u, nc);
    memcpy(e->frame.data() + ny + nc, v, nc);
    return PFV_OK;
}

static int host_entropy_staging(pfv_encoder *e)
{
    if (e->coef.resize((size_t)e->total_blocks * 256) && e->mv.resize((size_t)e->total_blocks * 2) && e->has.resize((size_t)e->total_blocks))
        return PFV_OK;
    return fail(e->ctx, PFV_ERR_NOMEM, "pinned staging for the host entropy path");
}

// One frame through the device entropy stage: planes up, kernels, payload size then payload bytes down.
static int encode_on_device(pfv_encoder *e, bool pframe)
{
    pfv_enc_session *s = e->hot;
    pfv_ctx *ctx = e->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_staging(s);
    if (!rc) rc = pfv_enc_entropy_enable(s, 0);
    if (rc) return rc;
    {   // the caller's planes are read until the first synchronisation below (pfv_enc_payload_sizes); every exit before it synchronises too
        const size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
        const bool packed = e->plane[1] == e->plane[0] + ny && e->plane[2] == e->plane[1] + nc;   // a packed frame: one copy
        hipError_t he = hipMemcpyAsync(s->st_frames, e->plane[0], packed ? ny + 2 * nc : ny, hipMemcpyHostToDevice, ctx->stream);
        if (!packed && he == hipSuccess) he = hipMemcpyAsync(s->st_frames + ny, e->plane[1], nc, hipMemcpyHostToDevice, ctx->stream);
        if (!packed && he == hipSuccess) he = hipMemcpyAsync(s->st_frames + ny + nc, e->plane[2], nc, hipMemcpyHostToDevice, ctx->stream);
        if (he != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return hip_fail(ctx, he, "plane upload"); }
    }
    rc = pframe ? pfv_enc_pframe_dev(s, s->st_frames, s->st_mv, s->st_has, s->st_coef) : pfv_enc_iframe_dev(s, s->st_frames, s->st_coef);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    // from here on prev_frame has moved to this frame: a failure leaves the encoder's reference ahead of the stream
    e->poisoned = true;
    rc = pframe ? pfv_enc_pack_pframe_dev(s, s->st_mv, s->st_has, s->st_coef) : pfv_enc_pack_iframe_dev(s, s->st_coef);
    uint32_t nbytes = 0;
    if (!rc) rc = pfv_enc_payload_sizes(s, &nbytes);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if (!e->payload.resize(std::max<size_t>(nbytes, 1 << 20))) return fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
    if ((rc = pfv_enc_payload_fetch(s, 0, e->payload.data(), nbytes))) return rc;
    e->poisoned = false;
    e->out.push_back(pframe ? 2 : 1);
    put_u32(e->out, nbytes);
    e->out.insert(e->out.end(), e->payload.data(), e->payload.data() + nbytes);
    return PFV_OK;
}

// 1 (default): RLE + Huffman + bit packing on the device; 0: on the host (serialize_iframe / serialize_pframe).  The
// bytes written are the same either way.
PFV_API int pfv_encoder_set_device_entropy(pfv_encoder *e, int on)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    e->device_entropy = on != 0;
    return PFV_OK;
}

// Encoder::encode_iframe (src/enc.rs:75-123)
PFV_API int pfv_encoder_encode_iframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    int rc = pack_frame(e, y, u, v);
    if (rc) return rc;
    if (e->device_entropy) return encode_on_device(e, false);      // an i-frame replaces prev_frame entirely: clears a poisoned state
    if ((rc = host_entropy_staging(e))) return rc;
    if ((rc = pfv_enc_iframe(e->hot, e->frame.data(), e->coef.data()))) return rc;
    e->poisoned = true;
    std::vector<uint8_t> payload;
    if (!serialize_iframe(payload, e->coef.data(), e->total_blocks))
        return fail(e->ctx, PFV_ERR_FORMAT, "coefficient needs more than 15 size bits (src/rle.rs:44)");
    put_packet(e->out, 1, &payload);
    e->poisoned = false;
    return PFV_OK;
}
// Encoder::encode_pframe (src/enc.rs:125-173)
PFV_API int pfv_encoder_encode_pframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    int rc = pack_frame(e, y, u, v);
    if (rc) return rc;
    // a previous frame failed after the encoder's reference had advanced but before its packet was written: a p-frame
    // now would predict from a frame the decoder never saw (the reference panics in that situation and the Encoder is gone)
    if (e->poisoned) return fail(e->ctx, PFV_ERR_STATE, "the previous frame failed after prev_frame had advanced: encode an i-frame next");
    if (e->device_entropy) return encode_on_device(e, true);
    if ((rc = host_entropy_staging(e))) return rc;
    if ((rc = pfv_enc_pframe(e->hot, e->frame.data(), e->mv.data(), e->has.data(), e->coef.data()))) return rc;
    e->poisoned = true;
    std::vector<uint8_t> payload;
    if (!serialize_pframe(payload, e->mv.data(), e->has.data(), e->coef.data(), e->total_blocks))
        return fail(e->ctx, PFV_ERR_FORMAT, "coefficient needs more than 15 size bits (src/rle.rs:44)");
    put_packet(e->out, 2, &payload);
    e->poisoned = false;
    return PFV_OK;
}
// Encoder::encode_dropframe (src/enc.rs:175-180): an i-frame packet with an empty payload
PFV_API int pfv_encoder_encode_dropframe(pfv_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:176)");
    put_packet(e->out, 1, nullptr);
    return PFV_OK;
}
// Encoder::finish (src/enc.rs:182-188): EOF packet
PFV_API int pfv_encoder_finish(pfv_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:183)");
    e->finished = true;
    put_packet(e->out, 0, nullptr);
    return PFV_OK;
}
PFV_API int pfv_encoder_bytes(pfv_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_encoder_bytes: bad argument");
    *data = e->out.data();
    *len = e->out.size();
    return PFV_OK;
}
// The reference streams every packet to its writer and keeps nothing (src/enc.rs:190-235); so does this: the bytes produced
// since the last drain are handed over and forgotten, only the current packet is ever resident.
PFV_API int pfv_encoder_drain(pfv_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_encoder_drain: bad argument");
    e->drained.swap(e->out);
    e->out.clear();
    *data = e->drained.data();
    *len = e->drained.size();
    return PFV_OK;
}
// Drop for Encoder (src/enc.rs:28-34): finishes the stream if the caller did not
PFV_API void pfv_encoder_destroy(pfv_encoder *e)
{
    if (!e) return;
    pfv_enc_session_destroy(e->hot);
    delete e;
}
// ------------------------------------------------------------------ batch encoder: n streams, pipelined
// n independent streams of one geometry encoded together (the reference runs one Encoder per stream, src/enc.rs:12-26):
// per frame step ONE upload, one launch per stage for all streams, one download of all payloads.  The upload of step t runs
// on its own copy stream while the host collects step t-1 (payload download, packet assembly, writers) and before the
// kernels of step t are enqueued, so PCIe, the kernels and the host work of neighbouring steps overlap:
//     encode(t):  [copy stream] frames(t) -> HBM      [host] finish step t-1: payloads -> writers
//                 [main stream] wait upload(t); k_enc_*; k_ent_*       (returns without waiting for them)
// Every writer receives exactly the bytes an Encoder of its own would have written (packets arrive one step late; finish
// flushes).
struct pfv_batch_encoder {
    pfv_ctx *ctx = nullptr;
    pfv_enc_session *hot = nullptr;
    int n = 0, width = 0, height = 0;
    size_t frame_bytes = 0, total_blocks = 0;
    pfv_write_cb write = nullptr;
    void *user = nullptr;
    std::vector<std::vector<uint8_t>> kept;   // write == NULL: per-stream bytes until pfv_batch_encoder_take
    std::vector<std::vector<uint8_t>> taken;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    uint8_t *in_host[2] = {nullptr, nullptr};   // page-locked [n][frame_bytes], filled by the caller
    uint8_t *in_dev[2] = {nullptr, nullptr};
    int16_t *coef = nullptr;
    int8_t *mv = nullptr;
    uint8_t *has = nullptr;
    PinnedBuf<uint8_t> payloads;
    std::vector<uint32_t> sizes;
    std::vector<uint64_t> offsets;
    std::vector<uint8_t> packet;
    long step = 0;
    int pending = -1;          // packet type of the step whose kernels are in flight, -1: none
    bool finished = false, poisoned = false;
};

static void be_emit(pfv_batch_encoder *b, int stream, const uint8_t *data, size_t len)
{
    if (b->write) b->write(b->user, stream, data, len);
    else b->kept[(size_t)stream].insert(b->kept[(size_t)stream].end(), data, data + len);
}
// the step in flight: wait for it, fetch every payload with one copy, hand the packets to the writers
static int be_collect(pfv_batch_encoder *b)
{
    if (b->pending < 0) return PFV_OK;
    const int type = b->pending;
    b->pending = -1;
    int rc = pfv_enc_payloads_fetch(b->hot, b->payloads.data(), b->payloads.size(), b->sizes.data(), b->offsets.data());
    if (rc == PFV_ERR_NOMEM) {   // very dense content: retry with the worst-case landing zone
        const size_t worst = (size_t)b->n * ((pfv_payload_worst_case(b->width, b->height) + 15) & ~(size_t)15);
        if (b->payloads.size() < worst && b->payloads.resize(worst))
            rc = pfv_enc_payloads_fetch(b->hot, b->payloads.data(), b->payloads.size(), b->sizes.data(), b->offsets.data());
    }
    if (rc) { b->poisoned = true; return rc; }
    for (int s = 0; s < b->n; s++) {
        const uint32_t nbytes = b->sizes[(size_t)s];
        uint8_t head[5] = {(uint8_t)type, (uint8_t)nbytes, (uint8_t)(nbytes >> 8), (uint8_t)(nbytes >> 16), (uint8_t)(nbytes >> 24)};
        if (b->write) {   // packet header (src/enc.rs:301-305, :453-457) + payload as one write
            b->packet.assign(head, head + 5);
            b->packet.insert(b->packet.end(), b->payloads.data() + b->offsets[(size_t)s], b->payloads.data() + b->offsets[(size_t)s] + nbytes);
            b->write(b->user, s, b->packet.data(), b->packet.size());
        } else {
            be_emit(b, s, head, 5);
            be_emit(b, s, b->payloads.data() + b->offsets[(size_t)s], nbytes);
        }
    }
    return PFV_OK;
}

PFV_API void pfv_batch_encoder_destroy(pfv_batch_encoder *b)
{
    if (!b) return;
    pfv_ctx *ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
    for (int i = 0; i < 2; i++) {
        if (b->ev_up[i]) (void)hipEventDestroy(b->ev_up[i]);
        if (b->in_host[i]) (void)hipHostFree(b->in_host[i]);
        if (b->in_dev[i]) (void)hipFree(b->in_dev[i]);
    }
    if (b->coef) (void)hipFree(b->coef);
    if (b->mv) (void)hipFree(b->mv);
    if (b->has) (void)hipFree(b->has);
    if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    pfv_enc_session_destroy(b->hot);
    delete b;
}

PFV_API int pfv_batch_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, int n_streams, pfv_write_cb write,
                                     void *user, pfv_batch_encoder **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_encoder_create: bad argument");
    *out = nullptr;
    if (framerate < 0 || framerate > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "framerate must fit u16 (src/enc.rs:197)");
    pfv_enc_session *hot = nullptr;
    int rc = pfv_enc_session_create(ctx, width, height, quality, n_streams, &hot);
    if (rc) return rc;
    pfv_batch_encoder *b = new pfv_batch_encoder();
    b->ctx = ctx; b->hot = hot; b->n = n_streams; b->width = width; b->height = height;
    b->write = write; b->user = user;
    b->frame_bytes = pfv_frame_bytes(width, height);
    b->total_blocks = (size_t)pfv_total_blocks(width, height);
    b->sizes.assign((size_t)n_streams, 0);
    b->offsets.assign((size_t)n_streams, 0);
    if (!write) { b->kept.resize((size_t)n_streams); b->taken.resize((size_t)n_streams); }
    const size_t in_bytes = (size_t)n_streams * b->frame_bytes, nmb = (size_t)n_streams * b->total_blocks;
    hipError_t e = hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&b->ev_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&b->in_host[i], in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&b->in_dev[i], in_bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&b->coef, nmb * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&b->mv, nmb * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&b->has, nmb);
    if (e != hipSuccess) {
        rc = hip_fail(ctx, e, "pfv_batch_encoder_create");
        pfv_batch_encoder_destroy(b);
        return rc;
    }
    rc = pfv_enc_entropy_enable(hot, 0);
    // landing zone for one step's payloads: typical content needs a fraction of the worst case; it grows on demand
    if (!rc && !b->payloads.resize(std::max<size_t>(in_bytes, 1 << 20))) rc = fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
    if (rc) { pfv_batch_encoder_destroy(b); return rc; }
    // header (src/enc.rs:190-219): magic, version, geometry, the four q-tables -- to every writer
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], nullptr);
    std::vector<uint8_t> head;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    head.insert(head.end(), magic, magic + 8);
    put_u32(head, 211);
    put_u16(head, (unsigned)width); put_u16(head, (unsigned)height); put_u16(head, (unsigned)framerate);
    put_u16(head, 4);
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 64; i++) put_u16(head, (unsigned)q[t][i]);
    for (int s = 0; s < n_streams; s++) be_emit(b, s, head.data(), head.size());
    *out = b;
    return PFV_OK;
}

// the page-locked [n_streams][frame_bytes] array to fill for the NEXT encode call (two of them alternate)
PFV_API uint8_t *pfv_batch_encoder_frames(pfv_batch_encoder *b) { return b ? b->in_host[b->step & 1] : nullptr; }

PFV_API int pfv_batch_encoder_encode(pfv_batch_encoder *b, int pframe, const uint8_t *frames)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    pfv_ctx *ctx = b->ctx;
    if (b->finished) return fail(ctx, PFV_ERR_STATE, "batch encoder already finished (src/enc.rs:80)");
    if (pframe && b->poisoned) return fail(ctx, PFV_ERR_STATE, "a previous step failed after prev_frame had advanced: encode i-frames next");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)(b->step & 1);
    const uint8_t *src = frames ? frames : b->in_host[slot];
    // in_dev[slot] was last read by the kernels of step t-2, which the collect of step t-1's call has waited for
    HIP_TRY(ctx, hipMemcpyAsync(b->in_dev[slot], src, (size_t)b->n * b->frame_bytes, hipMemcpyHostToDevice, b->copy_stream));
    // From here on the copy engine may be reading the CALLER's buffer: whatever way this call ends, it returns only once
    // that read is over ("free again when the call returns", pfv_hip.h).
    struct UploadGuard {
        hipStream_t s;
        bool armed;
        ~UploadGuard() { if (armed) (void)hipStreamSynchronize(s); }
    } guard{b->copy_stream, frames != nullptr};
    HIP_TRY(ctx, hipEventRecord(b->ev_up[slot], b->copy_stream));
    int rc = be_collect(b);            // step t-1 -> writers, while the upload of step t is on the wire
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, b->ev_up[slot], 0));
    rc = pframe ? pfv_enc_pframe_dev(b->hot, b->in_dev[slot], b->mv, b->has, b->coef) : pfv_enc_iframe_dev(b->hot, b->in_dev[slot], b->coef);
    if (rc) return rc;
    b->poisoned = true;                // until this step's packets have been written
    rc = pframe ? pfv_enc_pack_pframe_dev(b->hot, b->mv, b->has, b->coef) : pfv_enc_pack_iframe_dev(b->hot, b->coef);
    if (rc) return rc;
    if (frames) {
        guard.armed = false;
        HIP_TRY(ctx, hipStreamSynchronize(b->copy_stream));   // the caller's buffer is free again when this returns
    }
    b->pending = pframe ? 2 : 1;
    b->poisoned = false;
    b->step++;
    return PFV_OK;
}
// packets of the step in flight -> writers (encode does this for the previous step by itself)
PFV_API int pfv_batch_encoder_flush(pfv_batch_encoder *b)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    return be_collect(b);
}
PFV_API int pfv_batch_encoder_finish(pfv_batch_encoder *b)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    if (b->finished) return fail(b->ctx, PFV_ERR_STATE, "batch encoder already finished (src/enc.rs:183)");
    int rc = be_collect(b);
    if (rc) return rc;
    b->finished = true;
    const uint8_t eof[5] = {0, 0, 0, 0, 0};                                    // src/enc.rs:221-227
    for (int s = 0; s < b->n; s++) be_emit(b, s, eof, 5);
    return PFV_OK;
}
// write == NULL at creation: the bytes produced for one stream since the last take (valid until the next call on `b`)
PFV_API int pfv_batch_encoder_take(pfv_batch_encoder *b, int stream, const uint8_t **data, size_t *len)
{
    if (!b || !data || !len || stream < 0 || stream >= b->n || b->write) return fail(b ? b->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_batch_encoder_take: bad argument");
    b->taken[(size_t)stream].swap(b->kept[(size_t)stream]);
    b->kept[(size_t)stream].clear();
    *data = b->taken[(size_t)stream].data();
    *len = b->taken[(size_t)stream].size();
    return PFV_OK;
}

// ------------------------------------------------------------------ batch decoder: n streams, pipelined
// n .pfv streams of one geometry and one packet-type pattern (e.g. what a pfv_batch_encoder wrote) decoded together.  The
// packets of a step are bit-parsed (src/dec.rs:226-296, 328-417) on a worker pool, one task per stream, into (index, value)
// lists in page-locked memory; ONE segmented scatter kernel reads the lists straight from host memory, ONE decode launch
// serves all streams, ONE copy brings the frames back.  The parse of step t+1 runs while the device works on step t.
struct BdSet {   // host staging of one step (two sets alternate)
    PinnedBuf<uint32_t> idx;
    PinnedBuf<int16_t> val;
    PinnedBuf<uint32_t> counts;
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    std::vector<int> rc;               // per stream: 0, kSinkFull, PFV_ERR_*
    std::vector<uint8_t> qidx;         // per stream x 3
    std::vector<const uint8_t *> payload;
    std::vector<size_t> len;
    int type = 0;                      // 0 EOF, 1 i-frames, 2 p-frames, 3 drop frames; negative: error found by the scanner
    // device-entropy form of the step (PFV_OPT_ENTROPY_DECODE): what entd_prepare leaves for the k_entd_* kernels, per stream
    bool dev_form = false;
    PinnedBuf<uint8_t> bytes;          // the payloads, 16-byte aligned starts
    PinnedBuf<EdPacket> pk;            // [n]
    PinnedBuf<uint2> groups;
    std::vector<uint8_t> host_parse;   // per stream: the host parser reads this packet
    size_t bytes_total = 0;
};
struct pfv_batch_decoder {
    DecEntd entd;                      // switches, shape and counters of the device entropy stage
    DecWindow win[2];                  // its device buffers, per staging set: [n] packets, [n][total_blocks] lists / headers / coefficients
    hipStream_t win_stream = nullptr;  // the window of step t + 1 runs here, under the decode and download of step t
    pfv_ctx *ctx = nullptr;
    pfv_dec_session *hot = nullptr;
    int n = 0, width = 0, height = 0, framerate = 0, n_qtables = 0;
    size_t total_blocks = 0, frame_bytes = 0, cap = 0;
    std::vector<const uint8_t *> data;
    std::vector<size_t> len, pos;
    BdSet set[2];
    PinnedBuf<int16_t> dense;          // fallback for steps whose lists overflow
    ListStage hp;                      // device-entropy steps: a packet the device stage left to the host parser
    PinnedBuf<uint8_t> frames[2];
    uint8_t *frames_dev = nullptr;
    long step = 0, dense_steps = 0;
    bool eof = false;
    // worker pool
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    BdSet *job = nullptr;
    int next = 0, done = 0, generation = 0;
    bool quit = false;
};

static void bd_parse_one(pfv_batch_decoder *b, BdSet *s, int k)
{
    const size_t tb = b->total_blocks;
    if (s->dev_form) {   // the device reads the run streams: table, q indices, block headers and the payload's copy here
        EdPacket &pk = s->pk.data()[k];
        const EntdPrep r = entd_prepare(s->payload[(size_t)k], (uint32_t)s->len[(size_t)k], s->type, tb, b->n_qtables, b->entd.sub_bits, pk, s->bytes.data() + pk.byte_off);
        s->rc[(size_t)k] = r.rc;
        s->host_parse[(size_t)k] = r.host_parse;
        memcpy(&s->qidx[(size_t)k * 3], r.qidx, 3);
        s->counts.data()[k] = 0;
        return;
    }
    SparseSink sink{s->idx.data() + (size_t)k * b->cap, s->val.data() + (size_t)k * b->cap, b->cap};
    sink.offset = (size_t)k * tb * 256;
    uint8_t *q = &s->qidx[(size_t)k * 3];
    int rc = s->type == 2 ? parse_pframe_to(s->payload[(size_t)k], s->len[(size_t)k], (int)tb, b->n_qtables, s->mv.data() + (size_t)k * tb * 2,
                                            s->has.data() + (size_t)k * tb, sink, q)
                          : parse_iframe_to(s->payload[(size_t)k], s->len[(size_t)k], (int)tb, b->n_qtables, sink, q);
    s->counts.data()[k] = (uint32_t)sink.n;
    s->rc[(size_t)k] = rc;
}
static void bd_worker(pfv_batch_decoder *b)
{
    std::unique_lock<std::mutex> lk(b->m);
    int seen = 0;
    for (;;) {
        b->cv_work.wait(lk, [&] { return b->quit || b->generation != seen; });
        if (b->quit) return;
        seen = b->generation;
        BdSet *s = b->job;
        while (s && b->next < b->n) {
            const int k = b->next++;
            lk.unlock();
            bd_parse_one(b, s, k);
            lk.lock();
            if (++b->done == b->n) b->cv_done.notify_all();
        }
    }
}
// next frame packet of every stream (unknown packet types are skipped, src/dec.rs:216-219); starts the parse on the pool
static void bd_scan_and_start(pfv_batch_decoder *b, BdSet *s)
{
    s->type = 0;
    s->dev_form = false;        // only a step of frame packets takes the device form (set below)
    int first = -1;
    bool all_empty = true, any_empty = false;
    for (int k = 0; k < b->n; k++) {
        const uint8_t *d = b->data[(size_t)k];
        size_t p = b->pos[(size_t)k];
        int typ;
        size_t n = 0;
        for (;;) {
            if (p + 5 > b->len[(size_t)k]) { s->type = PFV_ERR_IO; return; }
            typ = d[p];
            n = (size_t)d[p + 1] | ((size_t)d[p + 2] << 8) | ((size_t)d[p + 3] << 16) | ((size_t)d[p + 4] << 24);
            if (typ == 0) break;
            if (p + 5 + n > b->len[(size_t)k]) { s->type = PFV_ERR_IO; return; }
            p += 5 + n;
            if (typ == 1 || typ == 2) break;
        }
        b->pos[(size_t)k] = p;
        if (first < 0) first = typ;
        else if (typ != first) { s->type = PFV_ERR_FORMAT; return; }     // the streams' packet types diverge at this step
        s->payload[(size_t)k] = typ ? d + p - n : nullptr;
        s->len[(size_t)k] = n;
        all_empty = all_empty && n == 0;
        any_empty = any_empty || n == 0;
    }
    if (first == 0) { s->type = 0; return; }
    if (first == 1 && all_empty) { s->type = 3; return; }                 // drop frames (src/dec.rs:188-202)
    if (any_empty) { s->type = first == 2 ? PFV_ERR_IO : PFV_ERR_FORMAT; return; }   // empty p-frame packet: truncated read (:204-214)
    s->type = first;
    s->dev_form = false;
    if (b->entd.on) {
        size_t total = 0;
        bool big = b->entd.force;
        for (int k = 0; k < b->n; k++) {
            s->pk.data()[k].byte_off = total;
            s->pk.data()[k].frame_off = (unsigned long long)k;
            total += (s->len[(size_t)k] + 16 + 15) & ~(size_t)15;
            big = big || s->len[(size_t)k] >= kDecEntdMinBytes;
        }
        s->bytes_total = total;
        s->dev_form = big && total < (1ull << 32) && s->bytes.resize(total + 64 > s->bytes.size() ? total + total / 2 + 64 : total + 64);
    }
    std::lock_guard<std::mutex> lk(b->m);
    b->job = s; b->next = 0; b->done = 0; b->generation++;
    b->cv_work.notify_all();
}
static void bd_join(pfv_batch_decoder *b, BdSet *s)
{
    if (s->type != 1 && s->type != 2) return;
    std::unique_lock<std::mutex> lk(b->m);
    while (b->job == s && b->next < b->n) {       // the caller helps (and is the whole pool when there are no workers)
        const int k = b->next++;
        lk.unlock();
        bd_parse_one(b, s, k);
        lk.lock();
        ++b->done;
    }
    b->cv_done.wait(lk, [&] { return b->done >= b->n; });
    b->job = nullptr;
}

}  // extern "C"

// The window of step s on set w (all on the window stream): payloads, packet descriptors, block headers and lists up, coefficient arrays
// cleared, k_entd_*, statuses down.
static int bd_window_enqueue(pfv_batch_decoder *b, BdSet *s, DecWindow &w)
{
    pfv_ctx *ctx = b->ctx;
    DecEntd &v = b->entd;
    const size_t S = (size_t)b->n, tb = b->total_blocks;
    int mrc = entd_windows_make(ctx, v, b->win, &b->win_stream, S, tb);
    if (mrc) return mrc;
    hipStream_t st = b->win_stream;
    size_t total_sub = 0, n_groups = 0, hdr_total = 0;
    unsigned max_hdr = 0;
    for (size_t k = 0; k < S; k++) {
        EdPacket &pk = s->pk.data()[k];
        if (s->host_parse[k] || s->rc[k]) pk.n_sub = pk.hdr_wgs = 0;
        pk.sub_first = (uint32_t)total_sub;
        pk.grp_first = (uint32_t)n_groups;
        pk.hdr_first = (uint32_t)hdr_total;
        total_sub += pk.n_sub;
        n_groups += (pk.n_sub + kEdOwn - 1) / kEdOwn;
        hdr_total += pk.hdr_wgs;
        max_hdr = std::max(max_hdr, (unsigned)pk.hdr_wgs);
    }
    if (total_sub >= 0xffffffffull) return fail(ctx, PFV_ERR_NOMEM, "batch decoder: payloads too large for one step of the device entropy stage");
    if (!s->groups.resize(n_groups + 1)) return fail(ctx, PFV_ERR_NOMEM, "pinned staging");
    {
        size_t g = 0;
        for (size_t k = 0; k < S; k++)
            for (uint32_t blk = 0; blk * (uint32_t)kEdOwn < s->pk.data()[k].n_sub; blk++) s->groups.data()[g++] = make_uint2((unsigned)k, blk);
    }
    auto room = [&](auto **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return PFV_OK;
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }               // the set is idle: its last window was consumed and decoded
        need += need / 2;
        HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(**p)));
        *cap = need;
        return PFV_OK;
    };
    int rc;
    if ((rc = room(&w.bytes_dev, &w.bytes_cap, s->bytes_total + 64))) return rc;
    if ((rc = room(&w.groups_dev, &w.groups_cap, n_groups + 1))) return rc;
    if ((rc = room(&w.sub_dev, &w.sub_cap, (total_sub + 1) * 4))) return rc;
    if ((rc = room(&w.wgsum_dev, &w.wgsum_cap, n_groups + 1))) return rc;
    if ((rc = room(&w.hdr_maps_dev, &w.hdr_maps_cap, (hdr_total + 1) * 8))) return rc;
    if ((rc = room(&w.hdr_start_dev, &w.hdr_start_cap, hdr_total + 1))) return rc;
    {   // every packet's list: its place in the window's pool from the packet's size
        size_t total = 0;
        w.list_room.assign(S, 0);
        for (size_t k = 0; k < S; k++) { w.list_room[k] = entd_pool_cap(tb, s->len[k]); total += w.list_room[k]; }
        w.lists.drop_spill();
        if ((rc = w.lists.room(ctx, total))) return rc;
        total = 0;
        for (size_t k = 0; k < S; k++) { w.lists.ptr_host.data()[k] = w.lists.ent + total; total += w.list_room[k]; }
        HIP_TRY(ctx, hipMemcpyAsync(w.lists.ptr_dev, w.lists.ptr_host.data(), S * sizeof(uint32_t *), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.bytes_dev, s->bytes.data(), s->bytes_total, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.pk_dev, s->pk.data(), S * sizeof(EdPacket), hipMemcpyHostToDevice, st));
    if (n_groups) HIP_TRY(ctx, hipMemcpyAsync(w.groups_dev, s->groups.data(), n_groups * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(w.status_dev, 0, S * sizeof(uint32_t), st));
    if (n_groups) {
        const size_t ts = w.sub_cap / 4;
        EdBufs eb{w.bytes_dev, w.pk_dev, w.groups_dev, w.sub_dev, w.sub_dev + ts, w.sub_dev + 2 * ts, w.wgsum_dev, w.coded_dev, w.lists.ptr_dev, w.lists.counts_dev, w.status_dev, 0u, 0u,
                  w.hdr_maps_dev, w.hdr_start_dev, w.mv_dev, w.has_dev};
        entd_launch(st, eb, (unsigned)S, (unsigned)n_groups, max_hdr, v.launches, v.inner);
        if ((rc = launch_check(ctx, "k_entd_*"))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.status_host.data(), w.status_dev, S * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(w.done, st));
    w.owner = (DecEvent *)s;       // an identity only: which staging set this window belongs to
    return PFV_OK;
}

extern "C" {
PFV_API void pfv_batch_decoder_destroy(pfv_batch_decoder *b)
{
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->m);
        b->quit = true;
        b->cv_work.notify_all();
    }
    for (auto &t : b->workers) t.join();
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->frames_dev) (void)hipFree(b->frames_dev);
    if (b->win_stream) { (void)hipStreamSynchronize(b->win_stream); (void)hipStreamDestroy(b->win_stream); }
    for (DecWindow &w : b->win) w.destroy();
    pfv_dec_session_destroy(b->hot);
    delete b;
}

PFV_API int pfv_batch_decoder_create(pfv_ctx *ctx, const uint8_t *const *streams, const size_t *lens, int n_streams, int n_threads,
                                     pfv_batch_decoder **out)
{
    if (!ctx || !streams || !lens || !out || n_streams <= 0 || n_threads < 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_decoder_create: bad argument");
    *out = nullptr;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    const uint8_t *d0 = streams[0];
    if (!d0 || lens[0] < 8) return fail(ctx, PFV_ERR_IO, "stream shorter than the magic");
    if (memcmp(d0, magic, 8) != 0) return fail(ctx, PFV_ERR_FORMAT, "bad magic (src/dec.rs:50-52)");
    if (lens[0] < 20) return fail(ctx, PFV_ERR_IO, "truncated header");
    const uint32_t ver = (uint32_t)d0[8] | ((uint32_t)d0[9] << 8) | ((uint32_t)d0[10] << 16) | ((uint32_t)d0[11] << 24);
    if (ver != 211) return fail(ctx, PFV_ERR_VERSION, "codec version is not 2.1.1 (src/dec.rs:57-59)");
    auto u16 = [&](size_t o) { return (int)d0[o] | ((int)d0[o + 1] << 8); };
    const int w = u16(12), h = u16(14), fps = u16(16), nq = u16(18);
    const size_t head = 20 + (size_t)nq * 128;
    for (int k = 0; k < n_streams; k++) {
        if (!streams[k] || lens[k] < head) return fail(ctx, PFV_ERR_IO, "truncated header");
        if (memcmp(streams[k], d0, head) != 0) return fail(ctx, PFV_ERR_FORMAT, "the streams must share one header (geometry, frame rate, q-tables)");
    }
    std::vector<int32_t> q((size_t)std::max(nq, 1) * 64, 1);
    for (int i = 0; i < nq * 64; i++) q[(size_t)i] = u16(20 + 2 * (size_t)i);
    // the coefficient lists address [stream][macroblock][256] with 32-bit flat indices (SparseSink, k_scatter_coef_seg)
    if (w > 0 && h > 0 && !(w & 1) && !(h & 1) && (uint64_t)n_streams * (uint64_t)pfv_total_blocks(w, h) * 256u > 0xffffffffull)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_decoder_create: n_streams x macroblocks x 256 exceeds the 32-bit coefficient index; use several batch decoders");
    pfv_dec_session *hot = nullptr;
    int rc = pfv_dec_session_create(ctx, w, h, q.data(), nq, n_streams, &hot);
    if (rc) return rc;
    pfv_batch_decoder *b = new pfv_batch_decoder();
    b->ctx = ctx; b->hot = hot; b->n = n_streams; b->width = w; b->height = h; b->framerate = fps; b->n_qtables = nq;
    b->total_blocks = (size_t)pfv_total_blocks(w, h);
    b->frame_bytes = pfv_frame_bytes(w, h);
    b->cap = b->total_blocks * 256 / 4;                       // per stream: denser than 1 in 4 -> dense fallback
    b->data.assign(streams, streams + n_streams);
    b->len.assign(lens, lens + n_streams);
    b->pos.assign((size_t)n_streams, head);
    const size_t S = (size_t)n_streams, tb = b->total_blocks;
    bool ok = true;
    for (auto &s : b->set) {
        ok = ok && s.idx.resize(S * b->cap) && s.val.resize(S * b->cap) && s.counts.resize(S) && s.mv.resize(S * tb * 2) && s.has.resize(S * tb);
        s.rc.assign(S, 0); s.qidx.assign(S * 3, 0); s.payload.assign(S, nullptr); s.len.assign(S, 0);
    }
    ok = ok && b->frames[0].resize(S * b->frame_bytes) && b->frames[1].resize(S * b->frame_bytes);
    if (ok && ctx->opt_entropy_decode != PFV_ENTROPY_DECODE_HOST && tb > 0) {   // the steps' run streams are read on the device (big payloads; every step under _DEVICE)
        DecEntd &v = b->entd;
        v.force = ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE;
        v.sub_bits = (uint32_t)ctx->opt_entdec_lane_bits; v.launches = ctx->opt_entdec_launches; v.inner = ctx->opt_entdec_inner;
        bool host_ok = true;                // the window stream and sets: with the first step that takes the device form (bd_window_enqueue)
        for (auto &s : b->set) {
            host_ok = host_ok && s.pk.resize(S);
            s.host_parse.assign(S, 0);
        }
        v.on = host_ok;
        if (!v.on && v.force) ok = false;
    }
    hipError_t e = ok ? hipMalloc((void **)&b->frames_dev, S * b->frame_bytes) : hipErrorOutOfMemory;
    if (e == hipSuccess && (rc = dec_staging(hot)) == PFV_OK) rc = pfv_dec_set_output_dev(hot, b->frames_dev);
    if (e != hipSuccess) rc = hip_fail(ctx, e, "pfv_batch_decoder_create");
    if (rc) { pfv_batch_decoder_destroy(b); return rc; }
    for (int t = 0; t < n_threads; t++) b->workers.emplace_back(bd_worker, b);
    bd_scan_and_start(b, &b->set[0]);                         // the first step is being parsed when create returns
    *out = b;
    return PFV_OK;
}
PFV_API int pfv_batch_decoder_width(const pfv_batch_decoder *b) { return b ? b->width : 0; }
PFV_API int pfv_batch_decoder_height(const pfv_batch_decoder *b) { return b ? b->height : 0; }
PFV_API int pfv_batch_decoder_framerate(const pfv_batch_decoder *b) { return b ? b->framerate : 0; }
// steps so far whose coefficient lists overflowed (denser than 1 non-zero in 4) and went up in the dense form
PFV_API long pfv_batch_decoder_dense_steps(const pfv_batch_decoder *b) { return b ? b->dense_steps : 0; }
PFV_API void pfv_batch_decoder_entropy_counts(const pfv_batch_decoder *b, long counts_out[2])
{
    if (!b || !counts_out) return;
    counts_out[0] = b->entd.packets_dev;
    counts_out[1] = b->entd.packets_host;
}

// One step for all streams: 1 = *frames_out points at [n_streams][frame_bytes] decoded frames (page-locked, valid until the
// call after next), 2 = a step of drop frames (no frames), 0 = end of the streams, negative = error (PFV_ERR_FORMAT also when
// the streams' packet types or q-table indices diverge).
PFV_API int pfv_batch_decoder_advance(pfv_batch_decoder *b, const uint8_t **frames_out)
{
    if (!b || !frames_out) return fail(b ? b->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_batch_decoder_advance: bad argument");
    pfv_ctx *ctx = b->ctx;
    *frames_out = nullptr;
    if (b->eof) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)(b->step & 1);
    BdSet *s = &b->set[slot];
    bd_join(b, s);
    if (s->type < 0) { b->eof = true; return fail(ctx, s->type, "batch decoder: truncated stream or diverging packet types"); }
    if (s->type == 0) { b->eof = true; return 0; }
    b->step++;
    if (s->type == 3) {
        bd_scan_and_start(b, &b->set[slot ^ 1]);
        return 2;
    }
    // From here on the step counter has advanced: an error exit that left the decoder usable would make the next call join a
    // slot whose contents are two steps old and decode it again as if it were new.  Every failure below ends the decoder
    // (b->eof), as the parse errors above do.
    const int rc_step = [&]() -> int {
    const size_t S = (size_t)b->n, tb = b->total_blocks;
    bool dense = false;
    for (size_t k = 0; k < S; k++) {
        if (s->rc[k] == kSinkFull) dense = true;
        else if (s->rc[k]) { b->eof = true; return fail(ctx, s->rc[k], "malformed packet payload"); }
        if (memcmp(&s->qidx[k * 3], &s->qidx[0], 3) != 0) { b->eof = true; return fail(ctx, PFV_ERR_FORMAT, "the streams use different q-table indices in this step"); }
    }
    pfv_dec_session *hot = b->hot;
    const size_t total = tb * S * 256;
    int rc = PFV_OK;
    if (s->dev_form) {   // the step's payloads through the device's entropy stage, the host parser for what it will not take
        DecEntd &v = b->entd;
        DecWindow &w = b->win[slot];
        if (w.owner != (DecEvent *)s && (rc = bd_window_enqueue(b, s, w))) return rc;     // not enqueued ahead (first step, or its headers were late)
        HIP_TRY(ctx, hipEventSynchronize(w.done));
        w.owner = nullptr;
        for (size_t k = 0; k < S; k++) {
            if (!s->host_parse[k] && !w.status_host.data()[k]) { v.packets_dev++; continue; }
            v.packets_host++;
            uint8_t q[3];
            const int prc = b->hp.parse(s->payload[k], s->len[k], s->type, tb, b->n_qtables, s->mv.data() + k * tb * 2, s->has.data() + k * tb, w.list_room[k], q);
            if (prc == PFV_ERR_NOMEM) return fail(ctx, prc, "pinned list staging");
            if (prc) { b->eof = true; return fail(ctx, prc, "malformed packet payload"); }
            if ((rc = upload_lists(ctx, w.lists, k, w.list_room[k], b->hp.ent.data(), b->hp.n, b->hp.counts.data(), ctx->stream))) return rc;
            if (s->type == 2) {     // its block headers with it (the device's read of them is not what is decoded)
                HIP_TRY(ctx, hipMemcpyAsync(w.mv_dev + k * tb * 2, s->mv.data() + k * tb * 2, tb * 2, hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(w.has_dev + k * tb, s->has.data() + k * tb, tb, hipMemcpyHostToDevice, ctx->stream));
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                 // the one staging list is used again
        }
        rc = dec_step(hot, s->type == 2, w.mv_dev, w.has_dev, w.lists.coefs(), &s->qidx[0]);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(b->frames[slot].data(), b->frames_dev, S * b->frame_bytes, hipMemcpyDeviceToHost, ctx->stream));
        BdSet *nx = &b->set[slot ^ 1];
        bd_scan_and_start(b, nx);
        if (nx->dev_form) {        // its headers now (the pool and this thread, while the frames above travel), then its window on the second stream
            bd_join(b, nx);
            bool sound = true;
            for (size_t k = 0; k < S; k++) sound = sound && !nx->rc[k];
            if (sound && bd_window_enqueue(b, nx, b->win[slot ^ 1]) != PFV_OK) b->win[slot ^ 1].owner = nullptr;   // tried again when its turn comes
        }
        if ((rc = pfv_dec_check(hot))) return rc;      // synchronises; bad-motion-vector flag (src/common.rs:258-259)
        *frames_out = b->frames[slot].data();
        return 1;
    } else {
    const bool lists_on_device_bus = s->idx.pinned && s->val.pinned && s->counts.pinned;   // page-locked: the kernel can read them
    if (!dense && !lists_on_device_bus) {   // pageable staging (locked-memory limit): expand the lists on the host instead
        if (!b->dense.resize(total)) return fail(ctx, PFV_ERR_NOMEM, "dense staging");
        memset(b->dense.data(), 0, total * 2);
        for (size_t k = 0; k < S; k++)
            for (uint32_t i = 0; i < s->counts.data()[k]; i++) b->dense.data()[s->idx.data()[k * b->cap + i]] = s->val.data()[k * b->cap + i];
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_coef, b->dense.data(), total * 2, hipMemcpyHostToDevice, ctx->stream));
    } else if (dense) {   // some list overflowed (very dense content): parse every stream into the dense form on this thread
        b->dense_steps++;
        if (!b->dense.resize(total)) return fail(ctx, PFV_ERR_NOMEM, "pinned dense staging");
        memset(b->dense.data(), 0, total * 2);
        for (size_t k = 0; k < S && !rc; k++) {
            DenseSink sink{b->dense.data() + k * tb * 256};
            uint8_t q[3];
            rc = s->type == 2 ? parse_pframe_to(s->payload[k], s->len[k], (int)tb, b->n_qtables, s->mv.data() + k * tb * 2, s->has.data() + k * tb, sink, q)
                              : parse_iframe_to(s->payload[k], s->len[k], (int)tb, b->n_qtables, sink, q);
        }
        if (rc) { b->eof = true; return fail(ctx, rc, "malformed packet payload"); }
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_coef, b->dense.data(), total * 2, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIP_TRY(ctx, hipMemsetAsync(hot->st_coef, 0, total * 2, ctx->stream));
        hipLaunchKernelGGL(k_scatter_coef_seg, dim3(64, (unsigned)S), dim3(kThreads), 0, ctx->stream, s->idx.data(), s->val.data(),
                           s->counts.data(), (uint32_t)b->cap, (uint32_t)total, hot->st_coef);
        if ((rc = launch_check(ctx, "k_scatter_coef_seg"))) return rc;
    }
    }
    if (s->type == 2) {
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_mv, s->mv.data(), S * tb * 2, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_has, s->has.data(), S * tb, hipMemcpyHostToDevice, ctx->stream));
        rc = pfv_dec_pframe_dev(hot, hot->st_mv, hot->st_has, hot->st_coef, &s->qidx[0]);
    } else {
        rc = pfv_dec_iframe_dev(hot, hot->st_coef, &s->qidx[0]);
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(b->frames[slot].data(), b->frames_dev, S * b->frame_bytes, hipMemcpyDeviceToHost, ctx->stream));
    bd_scan_and_start(b, &b->set[slot ^ 1]);       // parse of step t+1 under the device work of step t
    if ((rc = pfv_dec_check(hot))) return rc;      // synchronises; bad-motion-vector flag (src/common.rs:258-259)
    *frames_out = b->frames[slot].data();
    return 1;
    }();
    if (rc_step < 0) b->eof = true;
    return rc_step;
}

// payload serialisers alone (for tests: product vs oracle on identical coefficient input)
PFV_API size_t pfv_serialize_iframe_payload(const int16_t *coef, int total_blocks, uint8_t *out, size_t cap)
{
    std::vector<uint8_t> p;
    if (!coef || total_blocks <= 0 || !serialize_iframe(p, coef, total_blocks)) return 0;
    if (out && p.size() <= cap) memcpy(out, p.data(), p.size());
    return p.size();
}
PFV_API size_t pfv_serialize_pframe_payload(const int8_t *mv, const uint8_t *has_coef, const int16_t *coef, int total_blocks,
                                            uint8_t *out, size_t cap)
{
    std::vector<uint8_t> p;
    if (!mv || !has_coef || !coef || total_blocks <= 0 || !serialize_pframe(p, mv, has_coef, coef, total_blocks)) return 0;
    if (out && p.size() <= cap) memcpy(out, p.data(), p.size());
    return p.size();
}

// payload parsers alone (decode_iframe / decode_pframe up to the plane decode, src/dec.rs:226-296, 328-417); host only.
// coef_out: [total_blocks][256], zero-filled first.  Returns PFV_OK, PFV_ERR_FORMAT or PFV_ERR_IO.
PFV_API int pfv_parse_iframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int16_t *coef_out,
                                     uint8_t qidx_out[3])
{
    if (!payload || !coef_out || !qidx_out || total_blocks <= 0) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_iframe_payload: bad argument");
    int rc = parse_iframe(payload, len, total_blocks, n_qtables, coef_out, qidx_out);
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}
PFV_API int pfv_parse_pframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int8_t *mv_out,
                                     uint8_t *has_coef_out, int16_t *coef_out, uint8_t qidx_out[3])
{
    if (!payload || !mv_out || !has_coef_out || !coef_out || !qidx_out || total_blocks <= 0)
        return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_pframe_payload: bad argument");
    int rc = parse_pframe(payload, len, total_blocks, n_qtables, mv_out, has_coef_out, coef_out, qidx_out);
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}
// The sparse form the stream decoder uploads: up to `cap` (flat index, value) pairs; *n_out = pairs written.  Returns 1 when
// the list would overflow (the caller then parses the dense form).
PFV_API int pfv_parse_payload_sparse(int is_pframe, const uint8_t *payload, size_t len, int total_blocks, int n_qtables,
                                     int8_t *mv_out, uint8_t *has_coef_out, uint32_t *idx_out, int16_t *val_out, size_t cap,
                                     size_t *n_out, uint8_t qidx_out[3])
{
    if (!payload || !idx_out || !val_out || !n_out || !qidx_out || total_blocks <= 0 || (is_pframe && (!mv_out || !has_coef_out)))
        return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_payload_sparse: bad argument");
    SparseSink sink{idx_out, val_out, cap};
    int rc = is_pframe ? parse_pframe_to(payload, len, total_blocks, n_qtables, mv_out, has_coef_out, sink, qidx_out)
                       : parse_iframe_to(payload, len, total_blocks, n_qtables, sink, qidx_out);
    *n_out = sink.n;
    if (rc == kSinkFull) return 1;
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}

// Decoder::new (src/dec.rs:38-134).  `data` must stay valid for the decoder's lifetime (R: Read + Seek).
PFV_API int pfv_decoder_create(pfv_ctx *ctx, const uint8_t *data, size_t len, pfv_decoder **out)
{
    if (!ctx || !data || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_decoder_create: bad argument");
    *out = nullptr;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    if (len < 8) return fail(ctx, PFV_ERR_IO, "stream shorter than the magic (DecodeError::IOError)");
    if (memcmp(data, magic, 8) != 0) return fail(ctx, PFV_ERR_FORMAT, "bad magic (DecodeError::FormatError, src/dec.rs:50-52)");
    if (len < 12) return fail(ctx, PFV_ERR_IO, "truncated header");
    uint32_t ver = (uint32_t)data[8] | ((uint32_t)data[9] << 8) | ((uint32_t)data[10] << 16) | ((uint32_t)data[11] << 24);
    if (ver != 211) return fail(ctx, PFV_ERR_VERSION, "codec version is not 2.1.1 (DecodeError::VersionError, src/dec.rs:57-59)");
    if (len < 20) return fail(ctx, PFV_ERR_IO, "truncated header");
    auto u16 = [&](size_t o) { return (int)data[o] | ((int)data[o + 1] << 8); };
    int w = u16(12), h = u16(14), fps = u16(16), nq = u16(18);
    if (len < 20 + (size_t)nq * 128) return fail(ctx, PFV_ERR_IO, "truncated q-tables");
    std::vector<int32_t> q((size_t)std::max(nq, 1) * 64, 1);
    for (int i = 0; i < nq * 64; i++) q[i] = u16(20 + 2 * (size_t)i);
    pfv_dec_session *hot = nullptr;
    int rc = pfv_dec_session_create(ctx, w, h, q.data(), nq, 1, &hot);
    if (rc) return rc;
    pfv_decoder *d = new pfv_decoder();
    d->ctx = ctx; d->hot = hot; d->data = data; d->len = len;
    d->pos = d->reset_pos = d->scan_pos = 20 + (size_t)nq * 128;
    d->width = w; d->height = h; d->framerate = fps; d->n_qtables = nq;
    d->total_blocks = pfv_total_blocks(w, h);
    if (!d->retframe.resize(pfv_frame_bytes(w, h))) {
        pfv_decoder_destroy(d);
        return fail(ctx, PFV_ERR_NOMEM, "pfv_decoder_create: pinned staging");
    }
    memset(d->retframe.data(), 0, (size_t)w * h);                              // VideoFrame::new (frame.rs:12-26): Y 0, U/V 128
    memset(d->retframe.data() + (size_t)w * h, 128, d->retframe.size() - (size_t)w * h);
    if (ctx->opt_entropy_decode != PFV_ENTROPY_DECODE_HOST && d->total_blocks > 0) {   // the run streams of big packets are read on the device
        DecEntd &v = d->entd;
        v.force = ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE;
        v.sub_bits = (uint32_t)ctx->opt_entdec_lane_bits; v.launches = ctx->opt_entdec_launches; v.inner = ctx->opt_entdec_inner;
        v.on = true;                     // the window stream and sets: with the first packet that takes the device form (dec_window_enqueue)
    }
    const unsigned hw = std::thread::hardware_concurrency();
    if ((rc = pfv_decoder_set_lookahead(d, hw > 1 ? (int)std::min(4u, hw - 1) : 0))) {
        pfv_decoder_destroy(d);
        return rc;
    }
    *out = d;
    return PFV_OK;
}
}  // extern "C"

// ---- look-ahead machinery of pfv_decoder
static void dec_parse(pfv_decoder *d, DecEvent *e)   // any thread; touches only the event and the immutable stream
{
    const size_t tb = (size_t)d->total_blocks, cap = tb * 256 / 4;   // denser than 1 in 4: not worth a list
    e->dense = false;
    e->n_sparse = 0;
    e->dev_form = e->host_parse = false;
    if (d->entd.on && (d->entd.force || e->plen >= kDecEntdMinBytes)) {   // the device reads the run streams: only the headers here
        const uint32_t max_sub = (uint32_t)(((uint64_t)e->plen * 8 + d->entd.sub_bits - 1) / d->entd.sub_bits);
        if (!e->bytes.resize((size_t)e->plen + 32) || !e->pk.resize(1) || !e->groups.resize((size_t)max_sub / kEdOwn + 1) ||
            (e->type == 2 && (!e->mv.resize(tb * 2) || !e->has.resize(tb)))) {
            e->rc = PFV_ERR_NOMEM;
            return;
        }
        EdPacket &k = *e->pk.data();
        k.byte_off = 0; k.frame_off = 0;
        const EntdPrep r = entd_prepare(e->payload, e->plen, e->type, tb, d->n_qtables, d->entd.sub_bits, k, e->bytes.data());
        e->rc = r.rc;
        memcpy(e->qidx, r.qidx, 3);
        e->dev_form = true;
        e->host_parse = r.host_parse;
        if (r.rc || r.host_parse) k.n_sub = k.hdr_wgs = 0;
        const uint32_t ng = (k.n_sub + kEdOwn - 1) / kEdOwn;
        for (uint32_t g = 0; g < ng; g++) e->groups.data()[g] = make_uint2(0u, g);
        if (!e->rc && e->host_parse) {   // the host parser decides about this one, here, on this thread
            if (!e->coef.resize(tb * 256)) { e->rc = PFV_ERR_NOMEM; return; }
            e->rc = e->type == 1 ? parse_iframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->coef.data(), e->qidx)
                                 : parse_pframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(), e->coef.data(), e->qidx);
            e->dev_form = false;
            e->dense = true;
        }
        return;
    }
    if (!e->idx.resize(cap) || !e->val.resize(cap) || (e->type == 2 && (!e->mv.resize(tb * 2) || !e->has.resize(tb)))) {
        e->rc = PFV_ERR_NOMEM;
        return;
    }
    SparseSink sink{e->idx.data(), e->val.data(), cap};
    e->rc = e->type == 1 ? parse_iframe_to(e->payload, e->plen, d->total_blocks, d->n_qtables, sink, e->qidx)
                         : parse_pframe_to(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(), sink,
                                           e->qidx);
    e->n_sparse = sink.n;
    if (e->rc != kSinkFull) return;
    e->dense = true;
    if (!e->coef.resize(tb * 256)) {
        e->rc = PFV_ERR_NOMEM;
        return;
    }
    e->rc = e->type == 1 ? parse_iframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->coef.data(), e->qidx)
                         : parse_pframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(),
                                        e->coef.data(), e->qidx);
}
static void dec_worker(pfv_decoder *d)
{
    (void)hipSetDevice(d->ctx->device);   // the pinned landing zones are allocated from this thread
    std::unique_lock<std::mutex> lk(d->m);
    for (;;) {
        DecEvent *job = nullptr;
        for (size_t k = 0; k < d->count && !job; k++) {
            DecEvent *e = d->ring[(d->head + k) % d->ring.size()].get();
            if (e->state == DecEvent::QUEUED) job = e;
        }
        if (d->quit) return;
        if (!job) { d->cv_work.wait(lk); continue; }
        job->state = DecEvent::RUNNING;
        lk.unlock();
        dec_parse(d, job);
        lk.lock();
        job->state = DecEvent::DONE;
        d->cv_done.notify_all();
    }
}
// Walks packet headers from scan_pos exactly as the reference's loop would (dec.rs:174-222) and queues what it finds
// until the ring is full or an END / ERROR event is pending.  Caller holds the lock.
static void dec_scan(pfv_decoder *d)
{
    bool queued = false;
    while (d->count < d->ring.size() && !d->scan_stop) {
        DecEvent *e = d->ring[(d->head + d->count) % d->ring.size()].get();
        size_t pos = d->scan_pos;
        auto emit = [&](DecEvent::Kind kind, DecEvent::State st, size_t pos_after) {
            e->kind = kind; e->state = st; e->pos_after = pos_after;
            d->count++;
        };
        if (pos + 5 > d->len) {
            e->rc = PFV_ERR_IO; e->msg = "unexpected end of stream in a packet header";
            emit(DecEvent::ERROR, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        const uint8_t type = d->data[pos];
        const uint32_t plen = (uint32_t)d->data[pos + 1] | ((uint32_t)d->data[pos + 2] << 8) | ((uint32_t)d->data[pos + 3] << 16) |
                              ((uint32_t)d->data[pos + 4] << 24);
        pos += 5;
        if (type == 0) {   // EOF marker (:183-187)
            emit(DecEvent::END, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        if (pos + plen > d->len) {
            e->rc = PFV_ERR_IO; e->msg = "packet payload runs past the end of the stream";
            emit(DecEvent::ERROR, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        const uint8_t *payload = d->data + pos;
        pos += plen;
        d->scan_pos = pos;
        if (type != 1 && type != 2) continue;   // unknown packet: skipped (:216-219)
        if (type == 1 && plen == 0) {           // drop frame: nothing decoded, no callback (:190)
            emit(DecEvent::DROP, DecEvent::DONE, pos);
            continue;
        }
        e->type = type; e->payload = payload; e->plen = plen; e->rc = 0;
        emit(DecEvent::FRAME, DecEvent::QUEUED, pos);
        queued = true;
    }
    if (queued) d->cv_work.notify_all();
}
// Forget everything scanned ahead and continue from `pos`.  Caller holds the lock.
static void dec_rewind(pfv_decoder *d, std::unique_lock<std::mutex> &lk, size_t pos)
{
    for (;;) {   // a parse in flight keeps pointers into its event: let it finish
        bool running = false;
        for (auto &e : d->ring) running |= e->state == DecEvent::RUNNING;
        if (!running) break;
        d->cv_done.wait(lk);
    }
    if (d->win_stream) (void)hipStreamSynchronize(d->win_stream);          // a window enqueued ahead reads its event's buffers
    for (DecWindow &w : d->win) w.owner = nullptr;
    for (auto &e : d->ring) e->state = DecEvent::FREE;
    d->head = d->count = 0;
    d->scan_pos = d->pos = pos;
    d->scan_stop = false;
}
static void dec_stop_workers(pfv_decoder *d)
{
    {
        std::lock_guard<std::mutex> lk(d->m);
        d->quit = true;
    }
    d->cv_work.notify_all();
    for (auto &t : d->workers) t.join();
    d->workers.clear();
    d->quit = false;
}

extern "C" {

// Packets parsed ahead of the one being decoded, on `n_threads` worker threads (0: parse inline, no threads).  The
// default is min(4, hardware threads - 1).  Frames, order and error codes are those of the sequential loop.
PFV_API int pfv_decoder_set_lookahead(pfv_decoder *d, int n_threads)
{
    if (!d || n_threads < 0 || n_threads > 64) return fail(d ? d->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_decoder_set_lookahead: bad argument");
    dec_stop_workers(d);
    std::unique_lock<std::mutex> lk(d->m);
    dec_rewind(d, lk, d->pos);
    d->ring.clear();
    for (int i = 0; i < std::max(n_threads + 1, kDecWindows); i++) d->ring.emplace_back(new DecEvent());     // kDecWindows at least: the packets behind the current one are scanned (and, without threads, prepared by the caller's thread)
    lk.unlock();
    for (int i = 0; i < n_threads; i++) d->workers.emplace_back(dec_worker, d);
    return PFV_OK;
}

PFV_API void pfv_decoder_destroy(pfv_decoder *d)
{
    if (!d) return;
    dec_stop_workers(d);
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->win_stream) { (void)hipStreamSynchronize(d->win_stream); (void)hipStreamDestroy(d->win_stream); }
    if (d->frame_dev) (void)hipFree(d->frame_dev);
    for (DecWindow &w : d->win) w.destroy();
    pfv_dec_session_destroy(d->hot);
    delete d;
}
// on != 0: the decoded frame stays in device memory and the callback's y / u / v are DEVICE pointers to the packed frame (valid until the
// next advance call) -- for consumers on the GPU; the frame's download, more than half of a 4K call, is not paid
PFV_API int pfv_decoder_set_output_device(pfv_decoder *d, int on)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    pfv_ctx *ctx = d->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !d->frame_dev) HIP_TRY(ctx, hipMalloc((void **)&d->frame_dev, pfv_frame_bytes(d->width, d->height)));
    if (!on && d->frame_dev) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(d->frame_dev); d->frame_dev = nullptr; }
    return PFV_OK;
}
PFV_API void pfv_decoder_entropy_counts(const pfv_decoder *d, long counts_out[2])
{
    if (!d || !counts_out) return;
    counts_out[0] = d->entd.packets_dev;
    counts_out[1] = d->entd.packets_host;
}
PFV_API int pfv_decoder_width(const pfv_decoder *d) { return d ? d->width : 0; }          // dec.rs:136-138
PFV_API int pfv_decoder_height(const pfv_decoder *d) { return d ? d->height : 0; }        // dec.rs:140-142
PFV_API int pfv_decoder_framerate(const pfv_decoder *d) { return d ? d->framerate : 0; }  // dec.rs:144-146
// Decoder::reset (src/dec.rs:148-152)
PFV_API int pfv_decoder_reset(pfv_decoder *d)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    std::unique_lock<std::mutex> lk(d->m);
    d->eof = false;
    dec_rewind(d, lk, d->reset_pos);
    return PFV_OK;
}

}  // extern "C"

// The window of packet e on set w: uploads, cleared coefficient array, k_entd_*, status download -- all on the window stream.
static int dec_window_enqueue(pfv_decoder *d, DecEvent *e, DecWindow &w)
{
    pfv_ctx *ctx = d->ctx;
    DecEntd &v = d->entd;
    const size_t tb = (size_t)d->total_blocks;
    int mrc = entd_windows_make(ctx, v, d->win, &d->win_stream, 1, tb);
    if (mrc) return mrc;
    hipStream_t st = d->win_stream;
    const EdPacket &k = *e->pk.data();
    const uint32_t ng = (k.n_sub + kEdOwn - 1) / kEdOwn;
    auto room = [&](auto **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return PFV_OK;
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }           // the set is idle: its last window was consumed and decoded
        need += need / 2;
        HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(**p)));
        *cap = need;
        return PFV_OK;
    };
    int rc;
    if ((rc = room(&w.bytes_dev, &w.bytes_cap, (size_t)e->plen + 64))) return rc;
    if ((rc = room(&w.groups_dev, &w.groups_cap, (size_t)ng + 1))) return rc;
    if ((rc = room(&w.sub_dev, &w.sub_cap, ((size_t)k.n_sub + 1) * 4))) return rc;
    if ((rc = room(&w.wgsum_dev, &w.wgsum_cap, (size_t)ng + 1))) return rc;
    if ((rc = room(&w.hdr_maps_dev, &w.hdr_maps_cap, ((size_t)k.hdr_wgs + 1) * 8))) return rc;
    if ((rc = room(&w.hdr_start_dev, &w.hdr_start_cap, (size_t)k.hdr_wgs + 1))) return rc;
    w.list_room.assign(1, entd_pool_cap(tb, e->plen));
    w.lists.drop_spill();
    if ((rc = w.lists.room(ctx, w.list_room[0]))) return rc;
    w.lists.ptr_host.data()[0] = w.lists.ent;
    HIP_TRY(ctx, hipMemcpyAsync(w.lists.ptr_dev, w.lists.ptr_host.data(), sizeof(uint32_t *), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.bytes_dev, e->bytes.data(), (size_t)e->plen + 16, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.pk_dev, e->pk.data(), sizeof(EdPacket), hipMemcpyHostToDevice, st));
    if (ng) HIP_TRY(ctx, hipMemcpyAsync(w.groups_dev, e->groups.data(), ng * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(w.status_dev, 0, sizeof(uint32_t), st));
    if (ng) {
        const size_t ts = w.sub_cap / 4;
        EdBufs b{w.bytes_dev, w.pk_dev, w.groups_dev, w.sub_dev, w.sub_dev + ts, w.sub_dev + 2 * ts, w.wgsum_dev, w.coded_dev, w.lists.ptr_dev, w.lists.counts_dev, w.status_dev, 0u, 0u,
                 w.hdr_maps_dev, w.hdr_start_dev, w.mv_dev, w.has_dev};
        entd_launch(st, b, 1u, ng, k.hdr_wgs, v.launches, v.inner);
        if ((rc = launch_check(ctx, "k_entd_*"))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.status_host.data(), w.status_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(w.done, st));
    w.owner = e;
    return PFV_OK;
}
// One packet through the device's entropy stage, then the decode launch; before the frame is fetched, the window of the
// packet behind it -- if its headers are ready -- is put on the window stream, where it runs under this frame's decode and download.
static int dec_consume_entd(pfv_decoder *d, DecEvent *e)
{
    pfv_ctx *ctx = d->ctx;
    pfv_dec_session *hot = d->hot;
    DecEntd &v = d->entd;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t tb = (size_t)d->total_blocks;
    int rc;
    DecWindow *w = nullptr;
    for (DecWindow &x : d->win)
        if (x.owner == e) w = &x;
    if (!w) {                     // not enqueued ahead: now (every window is free or holds a packet behind this one; the last call ended with the stream idle)
        for (DecWindow &x : d->win)
            if (!w && !x.owner) w = &x;
        if (!w) { w = &d->win[0]; w->owner = nullptr; }
        if ((rc = dec_window_enqueue(d, e, *w))) { w->owner = nullptr; return rc; }
    }
    HIP_TRY(ctx, hipEventSynchronize(w->done));
    w->owner = nullptr;           // consumed (event objects are reused by the ring: a stale match would take this window for a later packet's)
    if (*w->status_host.data()) {   // the device stage is not certain about this payload: the host parser reads it and decides
        v.packets_host++;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // the staging list's last upload
        const int prc = d->hp.parse(e->payload, e->plen, e->type, tb, d->n_qtables, e->mv.data(), e->has.data(), w->list_room[0], e->qidx);
        if (prc) return fail(ctx, prc, prc == PFV_ERR_NOMEM ? "pinned list staging" : "malformed packet payload");
        if ((rc = upload_lists(ctx, w->lists, 0, w->list_room[0], d->hp.ent.data(), d->hp.n, d->hp.counts.data(), ctx->stream))) return rc;
        if (e->type == 2) {         // its block headers with it (the device's read of them is not what is decoded)
            HIP_TRY(ctx, hipMemcpyAsync(w->mv_dev, e->mv.data(), tb * 2, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(w->has_dev, e->has.data(), tb, hipMemcpyHostToDevice, ctx->stream));
        }
    } else {
        v.packets_dev++;
    }
    rc = dec_step(hot, e->type == 2, w->mv_dev, w->has_dev, w->lists.coefs(), e->qidx);
    if (rc) return rc;
    // The packets behind this one: their windows (payload upload, k_hdr_*, k_entd_*, status) go onto the window stream now, where they run
    // under this frame's decode and download -- up to kDecWindows - 1 of them: one packet's window is a chain of a dozen small kernels
    // (~0.15 ms of latency for a 4K p-frame), so with a single window ahead the chain of packet t + 1 only started when packet t's decode was
    // launched and every frame waited for most of it.  The window just consumed is still being read by the decode launched above: it is not
    // among the free ones until this call has returned.
    for (size_t k = 1; k < (size_t)kDecWindows; k++) {
        std::unique_lock<std::mutex> lk(d->m);
        DecEvent *nx = d->count > k ? d->ring[(d->head + k) % d->ring.size()].get() : nullptr;
        if (!nx) break;
        if (nx->state == DecEvent::QUEUED && d->workers.empty()) {
            // no parser threads (pfv_decoder_set_lookahead(d, 0)): this thread reads the packet's first 19 bytes and stages its payload now,
            // while the decode just launched runs -- with the block headers read on the device that is all a big packet needs from the host
            nx->state = DecEvent::RUNNING;
            lk.unlock();
            dec_parse(d, nx);
            lk.lock();
            nx->state = DecEvent::DONE;
        }
        const bool ready = nx->state == DecEvent::DONE && nx->kind == DecEvent::FRAME && nx->dev_form && !nx->rc;
        const bool stop = nx->state != DecEvent::DONE || nx->kind == DecEvent::END || nx->kind == DecEvent::ERROR;
        lk.unlock();
        if (stop) break;
        if (!ready) continue;
        DecWindow *free_w = nullptr;
        bool has = false;
        for (DecWindow &x : d->win) {
            has = has || x.owner == nx;
            if (!free_w && !x.owner && &x != w) free_w = &x;
        }
        if (has) continue;
        if (!free_w) break;
        if (dec_window_enqueue(d, nx, *free_w) != PFV_OK) { free_w->owner = nullptr; break; }   // it will be tried again when its turn comes
    }
    return pfv_dec_check(hot);
}

extern "C" {
// Decoder::advance_frame (src/dec.rs:169-224).  Returns 1 = Ok(true), 0 = Ok(false) (EOF), negative = error.
// onvideo(user, y, u, v, width, height) is called for every decoded frame (not for drop frames).
PFV_API int pfv_decoder_advance_frame(pfv_decoder *d, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    if (d->eof) return 0;
    std::unique_lock<std::mutex> lk(d->m);
    dec_scan(d);
    DecEvent *e = d->ring[d->head].get();
    while (e->state != DecEvent::DONE) {
        if (e->state == DecEvent::QUEUED) {   // nobody picked it up yet: parse it here
            e->state = DecEvent::RUNNING;
            lk.unlock();
            dec_parse(d, e);
            lk.lock();
            e->state = DecEvent::DONE;
        } else {
            d->cv_done.wait(lk);
        }
    }
    // consume the event; the slot stays reserved (FREE but not rescanned) until the device has read its buffers
    d->pos = e->pos_after;
    const DecEvent::Kind kind = e->kind;
    if (kind == DecEvent::END || kind == DecEvent::ERROR) {
        const int rc = e->rc;
        const char *msg = e->msg;
        dec_rewind(d, lk, d->pos);   // nothing was scanned past it; the next call rescans from pos like the reference
        if (kind == DecEvent::END) {
            d->eof = true;
            return 0;
        }
        return fail(d->ctx, rc, msg);
    }
    int rc = PFV_OK;
    if (kind == DecEvent::FRAME) {
        lk.unlock();   // workers keep parsing the packets behind this one while the device decodes it
        rc = e->rc;
        if (rc) rc = fail(d->ctx, rc, rc == PFV_ERR_NOMEM ? "pinned staging for a parsed packet" : "malformed packet payload");
        if (!rc && e->host_parse && !e->dev_form) d->entd.packets_host++;   // a packet of device size the host parser had to read (degenerate table, 64 MiB and more)
        if (!rc && e->dev_form) {
            rc = dec_consume_entd(d, e);
            if (rc && !d->entd.ready && !d->entd.force) {
                // PFV_ENTROPY_DECODE_AUTO and the window stream / sets could not be made (they are created with the first packet that takes the
                // device form): the device stage is switched off for this decoder and the host parser reads this packet -- and the ones the
                // parser threads have already prepared in device form, each when its turn comes.  An error only under PFV_ENTROPY_DECODE_DEVICE.
                (void)hipGetLastError();
                d->entd.on = false;
                dec_parse(d, e);
                rc = e->rc;
                if (rc) rc = fail(d->ctx, rc, rc == PFV_ERR_NOMEM ? "pinned staging for a parsed packet" : "malformed packet payload");
                else d->entd.packets_host++;
            }
        }
        if (rc || e->dev_form)
            ;
        else if (e->dense)
            rc = e->type == 1 ? pfv_dec_iframe(d->hot, e->coef.data(), e->qidx)
                              : pfv_dec_pframe(d->hot, e->mv.data(), e->has.data(), e->coef.data(), e->qidx);
        else
            rc = e->type == 1 ? pfv_dec_iframe_sparse(d->hot, e->idx.data(), e->val.data(), e->n_sparse, e->qidx)
                              : pfv_dec_pframe_sparse(d->hot, e->mv.data(), e->has.data(), e->idx.data(), e->val.data(),
                                                      e->n_sparse, e->qidx);
        if (!rc && d->frame_dev) {   // pfv_decoder_set_output_device: the retframe stays in device memory
            rc = pfv_dec_get_frame_dev(d->hot, d->frame_dev);
            if (!rc) rc = pfv_ctx_sync(d->ctx);
        } else if (!rc) {
            rc = pfv_dec_get_frame(d->hot, d->retframe.data());   // crop blits (:195-197, 209-211)
        }
        lk.lock();
    }
    e->state = DecEvent::FREE;
    d->head = (d->head + 1) % d->ring.size();
    d->count--;
    dec_scan(d);       // refill the freed slot right away
    lk.unlock();
    if (rc) return rc;
    if (kind == DecEvent::FRAME && onvideo) {
        size_t ny = (size_t)d->width * d->height, nc = (size_t)(d->width / 2) * (d->height / 2);
        const uint8_t *f = d->frame_dev ? d->frame_dev : d->retframe.data();
        onvideo(user, f, f + ny, f + ny + nc, d->width, d->height);
    }
    return 1;
}

// Decoder::advance_delta (src/dec.rs:154-167)
PFV_API int pfv_decoder_advance_delta(pfv_decoder *d, double delta, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    d->delta_accum += delta;
    double delta_per_frame = 1.0 / (double)d->framerate;
    while (d->delta_accum >= delta_per_frame) {
        int rc = pfv_decoder_advance_frame(d, onvideo, user);
        if (rc <= 0) return rc;
        d->delta_accum -= delta_per_frame;
    }
    return 1;
}

}  // extern "C"

// ------------------------------------------------------------------ device self-check of the encoders' f32 arithmetic (csrc/pfv_selfcheck.h)
int pfv_selfcheck_float_path(pfv_ctx *ctx, int part, uint64_t arg, uint64_t *checked, uint64_t *mismatches, int64_t first_bad[4])
{
    if (!ctx || !checked || !mismatches) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_selfcheck_float_path: bad argument");
    if (part < 0 || part > 6) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_selfcheck_float_path: part must be 0..6");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *checked = *mismatches = 0;
    ChkDev *res = nullptr;
    unsigned long long *n_cmp = nullptr;
    void *aux = nullptr, *aux2 = nullptr;
    auto cleanup = [&]() {
        if (res) (void)hipFree(res);
        if (n_cmp) (void)hipFree(n_cmp);
        if (aux) (void)hipFree(aux);
        if (aux2) (void)hipFree(aux2);
    };
#define CHK_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) { cleanup(); return hip_fail(ctx, e__, #expr); }         \
    } while (0)
    CHK_TRY(hipMalloc((void **)&res, sizeof(ChkDev)));
    CHK_TRY(hipMemset(res, 0, sizeof(ChkDev)));
    CHK_TRY(hipMalloc((void **)&n_cmp, sizeof(unsigned long long)));
    CHK_TRY(hipMemset(n_cmp, 0, sizeof(unsigned long long)));
    // parts 1, 2: 2^15 workgroups x 256 threads = the 2^23 values of |m|; a non-zero arg limits the workgroups (emulator runs)
    const unsigned m_blocks = (arg && arg < (1u << 15)) ? (unsigned)arg : (1u << 15);
    if (part == 0) {
        std::vector<float> rcp(65536);
        for (int q = 0; q < 65536; q++) rcp[q] = biased_rcp(q);
        CHK_TRY(hipMalloc(&aux, rcp.size() * sizeof(float)));
        CHK_TRY(hipMemcpy(aux, rcp.data(), rcp.size() * sizeof(float), hipMemcpyHostToDevice));
        const unsigned nq = arg ? (unsigned)std::min<uint64_t>(arg, 65535) : 65535u;      // arg: only the first `arg` values of q (emulator runs)
        hipLaunchKernelGGL(k_chk_quant_div, dim3(nq), dim3(256), 0, ctx->stream, (const float *)aux, kQuantMagic, res);
        *checked = (uint64_t)nq * 2 * 8193;
    } else if (part == 5) {
        hipLaunchKernelGGL(k_chk_residual, dim3(1), dim3(256), 0, ctx->stream, res);
        *checked = 2 * 256 * 256;
    } else if (part == 6) {
        const unsigned blocks = (arg && arg < (1u << 16)) ? (unsigned)arg : (1u << 16);   // 2^16 x 256 = 2^24 values of |x|
        hipLaunchKernelGGL(k_chk_iframe_pixel, dim3(blocks), dim3(256), 0, ctx->stream, res);
        *checked = (uint64_t)blocks * 256 * 2;
    } else if (part == 1) {
        hipLaunchKernelGGL(k_chk_quant_scale, dim3(m_blocks), dim3(256), 0, ctx->stream, res);
        *checked = (uint64_t)m_blocks * 256 * 2 * 10;
    } else if (part == 2) {
        // quantiser values spread over [1, 65535]: the small ones every quality table uses, powers of two and their neighbours, the u16 limit
        static const int kQs[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 20, 27, 40, 64, 83, 127, 128, 129, 1000, 4097, 32768, 65535};
        const int nq = (int)(sizeof kQs / sizeof kQs[0]);
        std::vector<float> rc(nq);
        for (int j = 0; j < nq; j++) rc[j] = biased_rcp(kQs[j]);
        CHK_TRY(hipMalloc(&aux, sizeof kQs));
        CHK_TRY(hipMemcpy(aux, kQs, sizeof kQs, hipMemcpyHostToDevice));
        CHK_TRY(hipMalloc(&aux2, nq * sizeof(float)));
        CHK_TRY(hipMemcpy(aux2, rc.data(), nq * sizeof(float), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chk_quant_pair, dim3(m_blocks), dim3(256), 0, ctx->stream, (const int *)aux, (const float *)aux2, nq, kQuantMagic, res);
        *checked = (uint64_t)m_blocks * 256 * 2 * 10 * nq;
    } else {
        // the four tables of every quality 0..10 (src/enc.rs:40-51), as the encoder sessions build them
        std::vector<ChkTab> tabs(44);
        for (int quality = 0; quality < 11; quality++) {
            int32_t q[4][64];
            float px_err;
            pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], &px_err);
            for (int k = 0; k < 4; k++) {
                ChkTab &T = tabs[quality * 4 + k];
                int rc = make_qtab(ctx, q[k], &T.qt);
                if (rc) { cleanup(); return rc; }
                for (int i = 0; i < 64; i++) {
                    T.q[i] = q[k][i];
                    T.cmax[i] = (int)enc_max_coef(q[k], (k < 2 ? 128.0 : 127.0) * 256.0, i);
                }
                if (!enc_float_exact(q[k], (k < 2 ? 128.0 : 127.0) * 256.0)) { cleanup(); return fail(ctx, PFV_ERR_STATE, "a quality table fails enc_float_exact"); }
            }
        }
        CHK_TRY(hipMalloc(&aux, tabs.size() * sizeof(ChkTab)));
        CHK_TRY(hipMemcpy(aux, tabs.data(), tabs.size() * sizeof(ChkTab), hipMemcpyHostToDevice));
        if (part == 3) {
            const unsigned n_pairs = (unsigned)std::min<uint64_t>(arg ? (arg + 1) / 2 : (1u << 19), 1u << 26);
            hipLaunchKernelGGL(k_chk_blocks, dim3((n_pairs + 63) / 64), dim3(64), 0, ctx->stream, (const ChkTab *)aux, 0x50465633ull, n_pairs, kQuantMagic, res, n_cmp);
        } else {
            const XformNorms &nm = xform_norms();
            CHK_TRY(hipMalloc(&aux2, 128));
            CHK_TRY(hipMemcpy(aux2, nm.fsign, 64, hipMemcpyHostToDevice));
            CHK_TRY(hipMemcpy((char *)aux2 + 64, nm.isign, 64, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_chk_worst_forward, dim3(4), dim3(64), 0, ctx->stream, (const ChkTab *)aux, (const signed char *)aux2, kQuantMagic, res, n_cmp);
            hipLaunchKernelGGL(k_chk_worst_inverse, dim3(2), dim3(64), 0, ctx->stream, (const ChkTab *)aux, (const signed char *)aux2 + 64, res, n_cmp);
        }
    }
    {
        int rc = launch_check(ctx, "pfv_selfcheck_float_path");
        if (rc) { cleanup(); return rc; }
    }
    CHK_TRY(hipStreamSynchronize(ctx->stream));
    ChkDev host;
    unsigned long long cmp = 0;
    CHK_TRY(hipMemcpy(&host, res, sizeof host, hipMemcpyDeviceToHost));
    CHK_TRY(hipMemcpy(&cmp, n_cmp, sizeof cmp, hipMemcpyDeviceToHost));
#undef CHK_TRY
    if (part == 3 || part == 4) *checked = cmp;
    *mismatches = host.mismatches;
    if (first_bad)
        for (int k = 0; k < 4; k++) first_bad[k] = host.first[k];
    cleanup();
    return PFV_OK;
}

#include "pfv_gop.hip"    // GOP-batched stream objects (pfv_gop_encoder, pfv_gop_decoder)

#include "pfv_comm.hip"   // multi-GPU control plane on RCCL (pfv_comm_*)

#include "pfv_prof_host.h"   // experiment builds only: fetch the phase timestamps (empty in the shipped build)
