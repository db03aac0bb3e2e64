// pfv_selfcheck.hip -- device self-check of the f32 arithmetic of the encoder kernels (csrc/pfv_selfcheck.h).
// Included by pfv_capi.hip after the kernels: the check kernels below call the very device functions k_enc_iframe<true> /
// k_enc_pframe<true> inline -- quant_scale, quant_div, quant_low16, ffdct8, fidct8 (pfv_kernels.hip) -- and compare them, on the
// GPU, with the integer arithmetic of the reference (src/dct.rs:88-99, 176-293; fdct8 / idct8 are the integer kernels' own
// butterflies, which the parity tests hold to the oracle bit for bit).  Nothing here is on the product path.
#include "pfv_selfcheck.h"

namespace pfv {

struct ChkDev {
    unsigned mismatches;      // saturating
    unsigned pad;
    long long first[4];       // operands + got + want of one failing evaluation
};
__device__ __forceinline__ void chk_report(ChkDev *r, long long a, long long b, long long got, long long want)
{
    const unsigned old = atomicAdd(&r->mismatches, 1u);
    if (old == 0u) { r->first[0] = a; r->first[1] = b; r->first[2] = got; r->first[3] = want; }
    if (old == 0xffffffffu) atomicAdd(&r->mismatches, 0xffffffffu);   // stay saturated
}

// part 0: quant_div + quant_low16, every n in [-8192, 8192] x every q in [1, 65535] (one workgroup per q; rcp[q] = QTab::rcp as
// make_qtab builds it, on the host)
__global__ __launch_bounds__(256) void k_chk_quant_div(const float *__restrict__ rcp, float magic, ChkDev *res)
{
    const int q = (int)blockIdx.x + 1;
    const float r = rcp[q];
    for (int t = (int)threadIdx.x; t <= 8192; t += 256) {
        const int n0 = t, n1 = -t;
        f2 biased;
        const f2 got = quant_div(f2{(float)n0, (float)n1}, r, magic, biased);
        const int w0 = n0 / q, w1 = n1 / q;                               // src/dct.rs:95: i32 `/` truncates toward zero
        if ((int)got[0] != w0 || quant_low16(biased[0]) != (int16_t)w0) chk_report(res, n0, q, (int)got[0], w0);
        if ((int)got[1] != w1 || quant_low16(biased[1]) != (int16_t)w1) chk_report(res, n1, q, (int)got[1], w1);
    }
}

// part 5: residual_f, every (source, prediction) byte pair -- thread = one source byte, loop over the predictions; the second
// half of the pair carries the negated delta
__global__ __launch_bounds__(256) void k_chk_residual(ChkDev *res)
{
    const int a = (int)threadIdx.x;
    for (int b = 0; b < 256; b++) {
        const f2 got = residual_f(f2{(float)a, (float)b}, f2{(float)b, (float)a});
        const int w0 = tdiv2(a - b) * 256, w1 = tdiv2(b - a) * 256;       // src/common.rs:118-119 (i16 delta), :304 (delta / 2 truncating, << 8)
        if (got[0] != (float)w0) chk_report(res, a, b, (long long)got[0], w0);
        if (got[1] != (float)w1) chk_report(res, b, a, (long long)got[1], w1);
    }
}

// part 6: the i-frame pixel, iframe_pixel_f + v_cvt_pk_u8_f32, for every inverse-transform output |x| < 2^24 (thread = (x, -x))
__global__ __launch_bounds__(256) void k_chk_iframe_pixel(ChkDev *res)
{
    const int x = (int)(blockIdx.x * 256u + threadIdx.x);                 // 0 .. 2^24 - 1
    const f2 p = iframe_pixel_f(f2{(float)x, (float)-x});
    const unsigned got = __builtin_amdgcn_cvt_pk_u8_f32(p[1], 1u, __builtin_amdgcn_cvt_pk_u8_f32(p[0], 0u, 0u));
    const int w0 = min(max((x >> 8) + 128, 0), 255), w1 = min(max((-x >> 8) + 128, 0), 255);     // src/common.rs:321
    if ((int)(got & 255u) != w0) chk_report(res, x, 0, got & 255u, w0);
    if ((int)((got >> 8) & 255u) != w1) chk_report(res, -x, 0, (got >> 8) & 255u, w1);
}

// the distinct values of DCT_SCALE_FACTOR (src/dct.rs:4-13)
__constant__ int kChkScales[10] = {22, 26, 28, 31, 32, 34, 35, 37, 39, 43};

// part 1: quant_scale, every |m| < 2^23 (thread = the pair (m, -m)) x every SCALE
__global__ __launch_bounds__(256) void k_chk_quant_scale(ChkDev *res)
{
    const int m = (int)(blockIdx.x * 256u + threadIdx.x);                 // 0 .. 2^23 - 1
    for (int k = 0; k < 10; k++) {
        const int S = kChkScales[k];
        const f2 got = quant_scale(f2{(float)m, (float)-m}, S << 16);
        const int w0 = (int)(((long long)m * S) >> 16), w1 = (int)(((long long)-m * S) >> 16);   // src/dct.rs:92: arithmetic shift
        if ((int)got[0] != w0) chk_report(res, m, S, (int)got[0], w0);
        if ((int)got[1] != w1) chk_report(res, -m, S, (int)got[1], w1);
    }
}

// part 2: the composed quantiser, every |m| < 2^23 x every SCALE x nq quantiser values (qs / rcps on the device)
__global__ __launch_bounds__(256) void k_chk_quant_pair(const int *__restrict__ qs, const float *__restrict__ rcps, int nq, float magic, ChkDev *res)
{
    const int m = (int)(blockIdx.x * 256u + threadIdx.x);
    for (int k = 0; k < 10; k++) {
        const int S = kChkScales[k];
        const int n0 = (int)(((long long)m * S) >> 16), n1 = (int)(((long long)-m * S) >> 16);
        for (int j = 0; j < nq; j++) {
            f2 biased;
            const f2 got = quant_pair_f(f2{(float)m, (float)-m}, S << 16, rcps[j], magic, biased);
            const int w0 = n0 / qs[j], w1 = n1 / qs[j];
            if ((int)got[0] != w0 || quant_low16(biased[0]) != (int16_t)w0) chk_report(res, m, ((long long)S << 32) | (unsigned)qs[j], (int)got[0], w0);
            if ((int)got[1] != w1 || quant_low16(biased[1]) != (int16_t)w1) chk_report(res, -m, ((long long)S << 32) | (unsigned)qs[j], (int)got[1], w1);
        }
    }
}

// One quantiser table as the check needs it: the integer table next to what the kernels read (QTab)
struct ChkTab {
    QTab qt;
    int q[64];        // raster order
    int cmax[64];     // largest coefficient magnitude the encoder can produce at each raster position (enc_float_exact's bound)
};

__device__ __forceinline__ unsigned chk_hash(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((x ^ (x >> 31)) >> 16);
}

// The closed loop of the encoders on a PAIR of 8x8 blocks held by one thread, integer form next to float form, every
// intermediate compared.  in[s][r * 8 + c]: 24.8 fixed-point samples ((px - 128) << 8 or (delta / 2) << 8, src/common.rs:291, :304).
// tabs[t0 .. t0 + nt): the tables to quantise with.  ident: what to report as operand a.  Returns values compared.
__device__ unsigned chk_closed_loop(const int (&in)[2][64], const ChkTab *tabs, int t0, int nt, float magic, ChkDev *res, long long ident)
{
    unsigned n_cmp = 0;
    int A[2][64];
    f2 X[64];
    for (int i = 0; i < 64; i++) {
        A[0][i] = in[0][i]; A[1][i] = in[1][i];
        X[i] = f2{(float)in[0][i], (float)in[1][i]};
    }
    // encode: rows, then columns (src/common.rs:294-295)
    for (int r = 0; r < 8; r++) {
        for (int s = 0; s < 2; s++) {
            int v[8];
            for (int k = 0; k < 8; k++) v[k] = A[s][r * 8 + k];
            fdct8(v);
            for (int k = 0; k < 8; k++) A[s][r * 8 + k] = v[k];
        }
        f2 x[8];
        for (int k = 0; k < 8; k++) x[k] = X[r * 8 + k];
        ffdct8(x);
        for (int k = 0; k < 8; k++) X[r * 8 + k] = x[k];
    }
    for (int c = 0; c < 8; c++) {
        for (int s = 0; s < 2; s++) {
            int v[8];
            for (int k = 0; k < 8; k++) v[k] = A[s][k * 8 + c];
            fdct8(v);
            for (int k = 0; k < 8; k++) A[s][k * 8 + c] = v[k];
        }
        f2 x[8];
        for (int k = 0; k < 8; k++) x[k] = X[k * 8 + c];
        ffdct8(x);
        for (int k = 0; k < 8; k++) X[k * 8 + c] = x[k];
    }
    for (int i = 0; i < 64; i++)
        for (int s = 0; s < 2; s++) {
            n_cmp++;
            if (X[i][s] != (float)A[s][i] || (int)X[i][s] != A[s][i]) chk_report(res, ident, 1000 + i, (long long)X[i][s], A[s][i]);
        }
    for (int t = t0; t < t0 + nt; t++) {
        const ChkTab &T = tabs[t];
        int D[2][64];
        f2 Y[64];
        // quantise (src/dct.rs:88-99, raster-indexed tables) and dequantise (src/dct.rs:75-86: QTab::deq is already permuted)
        for (int i = 0; i < 64; i++) {
            f2 biased;
            const f2 qf = quant_pair_f(X[i], kScale[i] << 16, T.qt.rcp[i], magic, biased);
            for (int s = 0; s < 2; s++) {
                const int n = (int)(((long long)A[s][i] * kScale[i]) >> 16);
                const int16_t c = (int16_t)(n / T.q[i]);
                n_cmp++;
                if ((int)qf[s] != (int)c || quant_low16(biased[s]) != c) chk_report(res, ident, ((long long)t << 32) | (2000 + i), (int)qf[s], c);
                D[s][i] = wmul((int)c, T.qt.deq[i]);
            }
            Y[i] = qf * f2s((float)T.qt.deq[i]);
        }
        // decode: columns, then rows (src/common.rs:315-316)
        for (int c = 0; c < 8; c++) {
            for (int s = 0; s < 2; s++) {
                int v[8];
                for (int k = 0; k < 8; k++) v[k] = D[s][k * 8 + c];
                idct8(v);
                for (int k = 0; k < 8; k++) D[s][k * 8 + c] = v[k];
            }
            f2 x[8];
            for (int k = 0; k < 8; k++) x[k] = Y[k * 8 + c];
            fidct8(x);
            for (int k = 0; k < 8; k++) Y[k * 8 + c] = x[k];
        }
        for (int r = 0; r < 8; r++) {
            for (int s = 0; s < 2; s++) {
                int v[8];
                for (int k = 0; k < 8; k++) v[k] = D[s][r * 8 + k];
                idct8(v);
                for (int k = 0; k < 8; k++) D[s][r * 8 + k] = v[k];
            }
            f2 x[8];
            for (int k = 0; k < 8; k++) x[k] = Y[r * 8 + k];
            fidct8(x);
            for (int k = 0; k < 8; k++) Y[r * 8 + k] = x[k];
        }
        for (int i = 0; i < 64; i++) {
            const f2 fl = f2floor(Y[i] * f2s(1.0f / 256.0f));             // inverse_half_f's (v >> 8)
            for (int s = 0; s < 2; s++) {
                n_cmp += 2;
                if (Y[i][s] != (float)D[s][i]) chk_report(res, ident, ((long long)t << 32) | (3000 + i), (long long)Y[i][s], D[s][i]);
                if (fl[s] != (float)(D[s][i] >> 8)) chk_report(res, ident, ((long long)t << 32) | (4000 + i), (long long)fl[s], D[s][i] >> 8);
            }
        }
    }
    return n_cmp;
}

// part 3: random blocks.  Thread = one pair of blocks; kind = pair & 3: 0 random pixels (i-frame input), 1 residual of two random
// pixel blocks, 2 full-swing 0 / 255 pixels, 3 full-swing +-255 residuals.  Tables: [quality][intra_l, intra_c, inter_l, inter_c];
// i-frame kinds run the 22 intra tables, residual kinds the 22 inter tables.
__global__ __launch_bounds__(64) void k_chk_blocks(const ChkTab *__restrict__ tabs, unsigned long long seed, unsigned n_pairs, float magic, ChkDev *res,
                                                   unsigned long long *n_cmp_out)
{
    const unsigned pair = blockIdx.x * 64u + threadIdx.x;
    if (pair >= n_pairs) return;
    const int kind = (int)(pair & 3u);
    int in[2][64];
    for (int s = 0; s < 2; s++)
        for (int i = 0; i < 64; i++) {
            const unsigned h = chk_hash(seed + ((unsigned long long)pair << 8) + (unsigned)(s * 64 + i));
            int v;
            if (kind == 0) v = ((int)(h & 255u) - 128) * 256;                                          // src/common.rs:291
            else if (kind == 1) v = tdiv2((int)(h & 255u) - (int)((h >> 8) & 255u)) * 256;             // src/common.rs:304
            else if (kind == 2) v = ((h & 1u) ? 127 : -128) * 256;
            else v = tdiv2((h & 1u) ? 255 : -255) * 256 * (((h >> 1) & 7u) ? 1 : 0);
            in[s][i] = v;
        }
    unsigned long long cmp = 0;
    for (int quality = 0; quality < 11; quality++)
        cmp += chk_closed_loop(in, tabs, quality * 4 + ((kind & 1) ? 2 : 0), 2, magic, res, pair);
    if (n_cmp_out) atomicAdd(n_cmp_out, cmp);
}

// part 4a: L1 worst cases of the forward transform -- the block A * sgn(F[u][r]) * sgn(F[v][c]) drives output (u, v) to its
// largest possible magnitude (and its negation to the most negative).  Thread = (u, v, sign, i-frame / residual amplitude).
// fsign[u * 8 + k] = sign of d out[u] / d in[k] of the 1-D forward transform.
__global__ __launch_bounds__(64) void k_chk_worst_forward(const ChkTab *__restrict__ tabs, const signed char *__restrict__ fsign, float magic, ChkDev *res,
                                                          unsigned long long *n_cmp_out)
{
    const int id = (int)(blockIdx.x * 64u + threadIdx.x);                 // 0 .. 255
    if (id >= 256) return;
    const int uv = id & 63, sgn = (id & 64) ? -1 : 1, resid = (id >> 7) & 1;
    const int u = uv >> 3, v = uv & 7;
    int in[2][64];
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            const int sg = sgn * fsign[u * 8 + r] * fsign[v * 8 + c];
            // i-frame samples span [-128, 127] << 8, residual samples [-127, 127] << 8; zero partial derivatives take the positive end
            in[0][r * 8 + c] = (resid ? 127 : (sg < 0 ? 128 : 127)) * (sg < 0 ? -256 : 256);
            in[1][r * 8 + c] = -in[0][r * 8 + c] == 32768 ? 127 * 256 : -in[0][r * 8 + c];   // the mirrored block in the other half of the pair
        }
    unsigned long long cmp = 0;
    for (int quality = 0; quality < 11; quality++)
        cmp += chk_closed_loop(in, tabs, quality * 4 + (resid ? 2 : 0), 2, magic, res, 1000000 + id);
    if (n_cmp_out) atomicAdd(n_cmp_out, cmp);
}

// part 4b: L1 worst cases of the inverse transform -- every coefficient at the largest magnitude the encoder can produce at
// its position (ChkTab::cmax), signed so that output pixel (x, y) is driven as far as it goes; both inverse passes in both
// forms.  Thread = (x, y, sign), loops over the 44 tables.  isign[x * 8 + k] = sign of d out[x] / d in[k] of the 1-D inverse.
__global__ __launch_bounds__(64) void k_chk_worst_inverse(const ChkTab *__restrict__ tabs, const signed char *__restrict__ isign, ChkDev *res,
                                                          unsigned long long *n_cmp_out)
{
    const int id = (int)(blockIdx.x * 64u + threadIdx.x);
    if (id >= 128) return;
    const int xy = id & 63, sgn = (id & 64) ? -1 : 1;
    const int x = xy >> 3, y = xy & 7;
    unsigned long long cmp = 0;
    for (int t = 0; t < 44; t++) {
        const ChkTab &T = tabs[t];
        int D[2][64];
        f2 Y[64];
        for (int i = 0; i < 64; i++) {
            const int u = i >> 3, v = i & 7;
            const int c = sgn * isign[x * 8 + u] * isign[y * 8 + v] * T.cmax[i];
            D[0][i] = wmul(c, T.qt.deq[i]);
            D[1][i] = wmul(-c, T.qt.deq[i]);
            Y[i] = f2{(float)c, (float)-c} * f2s((float)T.qt.deq[i]);
        }
        for (int c = 0; c < 8; c++) {
            for (int s = 0; s < 2; s++) {
                int w[8];
                for (int k = 0; k < 8; k++) w[k] = D[s][k * 8 + c];
                idct8(w);
                for (int k = 0; k < 8; k++) D[s][k * 8 + c] = w[k];
            }
            f2 f[8];
            for (int k = 0; k < 8; k++) f[k] = Y[k * 8 + c];
            fidct8(f);
            for (int k = 0; k < 8; k++) Y[k * 8 + c] = f[k];
        }
        for (int r = 0; r < 8; r++) {
            for (int s = 0; s < 2; s++) {
                int w[8];
                for (int k = 0; k < 8; k++) w[k] = D[s][r * 8 + k];
                idct8(w);
                for (int k = 0; k < 8; k++) D[s][r * 8 + k] = w[k];
            }
            f2 f[8];
            for (int k = 0; k < 8; k++) f[k] = Y[r * 8 + k];
            fidct8(f);
            for (int k = 0; k < 8; k++) Y[r * 8 + k] = f[k];
        }
        for (int i = 0; i < 64; i++) {
            const f2 fl = f2floor(Y[i] * f2s(1.0f / 256.0f));
            for (int s = 0; s < 2; s++) {
                cmp += 2;
                if (Y[i][s] != (float)D[s][i]) chk_report(res, 2000000 + id, ((long long)t << 32) | (3000 + i), (long long)Y[i][s], D[s][i]);
                if (fl[s] != (float)(D[s][i] >> 8)) chk_report(res, 2000000 + id, ((long long)t << 32) | (4000 + i), (long long)fl[s], D[s][i] >> 8);
            }
        }
    }
    if (n_cmp_out) atomicAdd(n_cmp_out, cmp);
}

}  // namespace pfv
