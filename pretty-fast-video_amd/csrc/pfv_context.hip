// pfv_context.hip -- context: pfv_ctx, options, timing events, HIP graphs (pfv_ctx_*, pfv_event_*, pfv_graph_*).
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
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
