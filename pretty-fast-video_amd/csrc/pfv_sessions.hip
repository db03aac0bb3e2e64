// pfv_sessions.hip -- encoder / decoder sessions: the hot-path state of enc::Encoder / dec::Decoder resident in HBM (pfv_enc_*, pfv_dec_*), the encoder session's device entropy stage.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
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
// slots_dev (pfv_gop_encoder with frames taken by reference): a device table of one frame pointer per slot, every pointer 16-byte aligned;
// slot k of the session reads the packed frame at slots_dev[k] instead of frames_dev + k * stride
static int enc_launch(pfv_enc_session *s, bool pframe, int first, int count, const uint8_t *frames_dev, int8_t *mv_dev, uint8_t *has_dev,
                      int16_t *coef_dev, const uint8_t *const *slots_dev = nullptr)
{
    pfv_ctx *ctx = s->ctx;
    const size_t stride = s->in_stride ? s->in_stride : (size_t)s->geom.src_frame_bytes;
    const uint8_t *src = frames_dev + (size_t)first * stride;
    const size_t mb0 = (size_t)first * (size_t)s->geom.mbs_per_frame, pad0 = (size_t)first * (size_t)s->geom.pad_frame_bytes;
    FrameGeom g = enc_win_geom(s, count, src);
    if (slots_dev) g.src_slots = slots_dev + first;
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
