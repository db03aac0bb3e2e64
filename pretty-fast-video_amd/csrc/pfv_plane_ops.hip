// pfv_plane_ops.hip -- the six plane-level operators on host buffers (src/common.rs:351-521), plane helpers on device planes, device memory helpers, geometry queries.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
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
