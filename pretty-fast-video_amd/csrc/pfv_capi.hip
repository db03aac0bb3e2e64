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

// the C ABI by object (one translation unit; the order is the dependency order)
#include "pfv_context.hip"
#include "pfv_launch.hip"
#include "pfv_plane_ops.hip"
#include "pfv_sessions.hip"
#include "pfv_stream_objects.hip"
#include "pfv_batch_objects.hip"
#include "pfv_decoder_object.hip"

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
