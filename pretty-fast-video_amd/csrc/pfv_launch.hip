// pfv_launch.hip -- internal helpers shared by the operators and the sessions: q-table preparation, frame geometry, kernel launchers.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
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
