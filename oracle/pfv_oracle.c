/*
 * pfv_oracle.c -- CPU restatement of the pfv-rs (Pretty Fast Video 0.2.2, codec 2.1.1)
 * per-macroblock transform / motion hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libpfv_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * Parity status: the reference is Rust and no Rust toolchain exists in the build image,
 * and every binary fixture of the reference is a Git-LFS pointer stub, so this oracle is
 * pinned by (a) line-by-line restatement of the cited reference source, (b) an
 * independently written numpy restatement (oracle/pfv_oracle_np.py) that must agree
 * bit-for-bit, and (c) the two inline test inputs of the reference's own unit tests
 * (src/lib.rs:38, src/lib.rs:61-66 -- those tests only print, they assert nothing).
 * => "parity unpinned by upstream golden vectors"; see DESIGN.md.
 *
 * Arithmetic conventions restated from Rust (release profile):
 *   - i32 add/sub/mul wrap (two's complement)          -> done in uint32_t here
 *   - `/` on i32 truncates toward zero                  -> C99 `/` does the same
 *   - `>>` on i32 is an arithmetic shift                -> sra() below
 *   - `as i16` / `as i8` wrap, `as u8` only after clamp
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

#define PFVO_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- constant tables */
/* src/dct.rs:1 */
#define FP_BITS 8
/* src/dct.rs:4-13 (data) */
static const int32_t DCT_SCALE_FACTOR[64] = {
    32, 37, 34, 26, 32, 26, 34, 37, 37, 43, 39, 31, 37, 31, 39, 43,
    34, 39, 35, 28, 34, 28, 35, 39, 26, 31, 28, 22, 26, 22, 28, 31,
    32, 37, 34, 26, 32, 26, 34, 37, 26, 31, 28, 22, 26, 22, 28, 31,
    34, 39, 35, 28, 34, 28, 35, 39, 37, 43, 39, 31, 37, 31, 39, 43,
};
/* src/dct.rs:16-25 (data) */
static const int32_t Q_TABLE_INTRA[64] = {
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83,
};
/* src/dct.rs:28-37 (data): all 16 */
#define Q_TABLE_INTER_VALUE 16
/* src/dct.rs:39-42 (data) */
static const uint8_t INV_ZIGZAG_TABLE[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
    3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63,
};
/* src/dct.rs:44-47 (data) */
static const uint8_t ZIGZAG_TABLE[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

PFVO_API void pfvo_tables(int32_t scale[64], int32_t q_intra[64], int32_t q_inter[64],
                          uint8_t inv_zigzag[64], uint8_t zigzag[64])
{
    for (int i = 0; i < 64; i++) {
        scale[i] = DCT_SCALE_FACTOR[i];
        q_intra[i] = Q_TABLE_INTRA[i];
        q_inter[i] = Q_TABLE_INTER_VALUE;
        inv_zigzag[i] = INV_ZIGZAG_TABLE[i];
        zigzag[i] = ZIGZAG_TABLE[i];
    }
}

/* ---------------------------------------------------------------- wrapping helpers */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t sra(int32_t a, int s)
{
    /* arithmetic shift right without relying on implementation-defined >> of negatives */
    return a >= 0 ? (a >> s) : ~((~a) >> s);
}

/* ---------------------------------------------------------------- 1-D transforms */
/* src/dct.rs:176-239  DctMatrix8x8::fdct */
PFVO_API void pfvo_fdct8(int32_t v[8])
{
    int32_t i0 = v[0], i1 = v[1], i2 = v[2], i3 = v[3], i4 = v[4], i5 = v[5], i6 = v[6], i7 = v[7];
    /* stage 1 (dct.rs:188-195) */
    int32_t a0 = wadd(i0, i7), a1 = wadd(i1, i6), a2 = wadd(i2, i5), a3 = wadd(i3, i4);
    int32_t a4 = wsub(i0, i7), a5 = wsub(i1, i6), a6 = wsub(i2, i5), a7 = wsub(i3, i4);
    /* even stage 2 (dct.rs:198-201) */
    int32_t b0 = wadd(a0, a3), b1 = wadd(a1, a2), b2 = wsub(a0, a3), b3 = wsub(a1, a2);
    /* even stage 3 (dct.rs:204-207) */
    int32_t c0 = wadd(b0, b1);
    int32_t c1 = wsub(b0, b1);
    int32_t c2 = wadd(wadd(b2, b2 / 4), b3 / 2);
    int32_t c3 = wsub(wsub(b2 / 2, b3), b3 / 4);
    /* odd stage 2 (dct.rs:211-214) */
    int32_t b4 = wsub(wadd(wadd(a7 / 4, a4), a4 / 4), a4 / 16);
    int32_t b7 = wadd(wsub(wsub(a4 / 4, a7), a7 / 4), a7 / 16);
    int32_t b5 = wsub(wsub(wadd(a5, a6), a6 / 4), a6 / 16);
    int32_t b6 = wadd(wadd(wsub(a6, a5), a5 / 4), a5 / 16);
    /* odd stage 3 (dct.rs:217-220) */
    int32_t c4 = wadd(b4, b5), c5 = wsub(b4, b5), c6 = wadd(b6, b7), c7 = wsub(b6, b7);
    /* odd stage 4 (dct.rs:223-226) */
    int32_t d4 = c4, d5 = wadd(c5, c7), d6 = wsub(c5, c7), d7 = c6;
    /* permute/output (dct.rs:229-236) */
    v[0] = c0; v[1] = d4; v[2] = c2; v[3] = d6; v[4] = c1; v[5] = d5; v[6] = c3; v[7] = d7;
}

/* src/dct.rs:241-293  DctMatrix8x8::idct */
PFVO_API void pfvo_idct8(int32_t v[8])
{
    /* input permutation (dct.rs:243-250) */
    int32_t c0 = v[0], d4 = v[1], c2 = v[2], d6 = v[3], c1 = v[4], d5 = v[5], c3 = v[6], d7 = v[7];
    /* odd stage 4 (dct.rs:253-256) */
    int32_t c4 = d4, c5 = wadd(d5, d6), c7 = wsub(d5, d6), c6 = d7;
    /* odd stage 3 (dct.rs:259-262) */
    int32_t b4 = wadd(c4, c5), b5 = wsub(c4, c5), b6 = wadd(c6, c7), b7 = wsub(c6, c7);
    /* even stage 3 (dct.rs:265-268) */
    int32_t b0 = wadd(c0, c1), b1 = wsub(c0, c1);
    int32_t b2 = wadd(wadd(c2, c2 / 4), c3 / 2);
    int32_t b3 = wsub(wsub(c2 / 2, c3), c3 / 4);
    /* odd stage 2 (dct.rs:271-274) */
    int32_t a4 = wsub(wadd(wadd(b7 / 4, b4), b4 / 4), b4 / 16);
    int32_t a7 = wadd(wsub(wsub(b4 / 4, b7), b7 / 4), b7 / 16);
    int32_t a5 = wadd(wadd(wsub(b5, b6), b6 / 4), b6 / 16);
    int32_t a6 = wsub(wsub(wadd(b6, b5), b5 / 4), b5 / 16);
    /* even stage 2 (dct.rs:277-280) */
    int32_t a0 = wadd(b0, b2), a1 = wadd(b1, b3), a2 = wsub(b1, b3), a3 = wsub(b0, b2);
    /* stage 1 (dct.rs:283-290) */
    v[0] = wadd(a0, a4); v[1] = wadd(a1, a5); v[2] = wadd(a2, a6); v[3] = wadd(a3, a7);
    v[4] = wsub(a3, a7); v[5] = wsub(a2, a6); v[6] = wsub(a1, a5); v[7] = wsub(a0, a4);
}

/* src/dct.rs:139-145 / :148-154 / :157-163 / :166-172 -- row / column drivers */
static void transform_rows(int32_t m[64], void (*f)(int32_t *))
{
    for (int r = 0; r < 8; r++) f(&m[r * 8]);
}
static void transform_cols(int32_t m[64], void (*f)(int32_t *))
{
    for (int c = 0; c < 8; c++) {
        int32_t col[8];
        for (int r = 0; r < 8; r++) col[r] = m[c + r * 8];
        f(col);
        for (int r = 0; r < 8; r++) m[c + r * 8] = col[r];
    }
}

/* src/dct.rs:88-99  DctMatrix8x8::encode -- scale, >>16, truncating /q, zigzag.
 * SCALE and q are indexed by the RASTER index. */
PFVO_API void pfvo_dct_encode(const int32_t m[64], const int32_t q[64], int16_t out[64])
{
    for (int i = 0; i < 64; i++) {
        int idx = ZIGZAG_TABLE[i];
        int32_t n = sra(wmul(m[idx], DCT_SCALE_FACTOR[idx]), FP_BITS * 2);
        int32_t d = q[idx];
        out[i] = (int16_t)(uint16_t)(uint32_t)(n / d);
    }
}

/* src/dct.rs:75-86  DctMatrix8x8::decode -- unzigzag, dequantise.
 * NOTE the asymmetry: SCALE and q are indexed by the ZIGZAG POSITION idx, not by the
 * raster index i.  Reproduced on purpose. */
PFVO_API void pfvo_dct_decode(const int16_t src[64], const int32_t q[64], int32_t m[64])
{
    for (int i = 0; i < 64; i++) {
        int idx = INV_ZIGZAG_TABLE[i];
        int32_t n = wmul((int32_t)src[idx], DCT_SCALE_FACTOR[idx]);
        int32_t d = q[idx];
        m[i] = wmul(n, d);
    }
}

/* src/common.rs:287-298  VideoPlane::encode_subblock (px: 8x8 raster, stride given) */
static void encode_subblock(const uint8_t *px, int stride, const int32_t q[64], int16_t out[64])
{
    int32_t m[64];
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++)
            m[r * 8 + c] = (int32_t)((uint32_t)((int32_t)px[r * stride + c] - 128) << FP_BITS);
    transform_rows(m, pfvo_fdct8);
    transform_cols(m, pfvo_fdct8);
    pfvo_dct_encode(m, q, out);
}

/* src/common.rs:300-311  VideoPlane::encode_subblock_delta (d: 8x8 i16, stride given) */
static void encode_subblock_delta(const int16_t *d, int stride, const int32_t q[64], int16_t out[64])
{
    int32_t m[64];
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++)
            m[r * 8 + c] = (int32_t)((uint32_t)((int32_t)d[r * stride + c] / 2) << FP_BITS);
    transform_rows(m, pfvo_fdct8);
    transform_cols(m, pfvo_fdct8);
    pfvo_dct_encode(m, q, out);
}

/* src/common.rs:313-325  VideoPlane::decode_subblock -- columns first, then rows */
static void decode_subblock(const int16_t in[64], const int32_t q[64], uint8_t out[64])
{
    int32_t m[64];
    pfvo_dct_decode(in, q, m);
    transform_cols(m, pfvo_idct8);
    transform_rows(m, pfvo_idct8);
    for (int i = 0; i < 64; i++) {
        int32_t v = wadd(sra(m[i], FP_BITS), 128);
        out[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

PFVO_API void pfvo_encode_subblock(const uint8_t px[64], const int32_t q[64], int16_t out[64])
{
    encode_subblock(px, 8, q, out);
}
PFVO_API void pfvo_encode_subblock_delta(const int16_t d[64], const int32_t q[64], int16_t out[64])
{
    encode_subblock_delta(d, 8, q, out);
}
PFVO_API void pfvo_decode_subblock(const int16_t in[64], const int32_t q[64], uint8_t out[64])
{
    decode_subblock(in, q, out);
}

/* ---------------------------------------------------------------- macroblock level */
/* src/common.rs:141-152  encode_block: quadrants (0,0) (8,0) (0,8) (8,8) */
static void encode_block(const uint8_t mb[256], const int32_t q[64], int16_t out[256])
{
    encode_subblock(mb + 0, 16, q, out + 0);
    encode_subblock(mb + 8, 16, q, out + 64);
    encode_subblock(mb + 8 * 16, 16, q, out + 128);
    encode_subblock(mb + 8 * 16 + 8, 16, q, out + 192);
}

/* src/common.rs:238-252 + :88-96  decode_block + blit_subblock */
static void decode_block(const int16_t in[256], const int32_t q[64], uint8_t mb[256])
{
    static const int ox[4] = {0, 8, 0, 8}, oy[4] = {0, 0, 8, 8};
    for (int s = 0; s < 4; s++) {
        uint8_t sb[64];
        decode_subblock(in + s * 64, q, sb);
        for (int r = 0; r < 8; r++) memcpy(mb + (oy[s] + r) * 16 + ox[s], sb + r * 8, 8);
    }
}

/* src/common.rs:125-139  calc_error: f32 SSD in raster order with early return */
static float calc_error(const uint8_t a[256], const uint8_t *b, int bstride, float ref_lms)
{
    float sum = 0.0f;
    for (int r = 0; r < 16; r++) {
        for (int c = 0; c < 16; c++) {
            float diff = (float)a[r * 16 + c] - (float)b[r * bstride + c];
            sum += diff * diff;
            if (sum >= ref_lms) return sum;
        }
    }
    return sum;
}

/* src/common.rs:154-204  block_search (recursion unrolled into a loop over stepsize;
 * every level re-evaluates its centre with ref_lms = +inf exactly as the reference). */
static void block_search(const uint8_t src[256], const uint8_t *ref, int refw, int refh, int cx, int cy,
                         int *out_dx, int *out_dy, float *out_err)
{
    int tot_dx = 0, tot_dy = 0;
    float best_err = INFINITY;
    for (int stepsize = 8; stepsize >= 1; stepsize /= 2) {
        int best_dx = 0, best_dy = 0;
        best_err = calc_error(src, ref + (size_t)cy * refw + cx, refw, INFINITY); /* :161-165 */
        for (int my = -1; my < 2; my++) {                                          /* :168 */
            int offsy = cy + my * stepsize;
            if (offsy < 0 || offsy > refh - 16) continue;                         /* :171 */
            for (int mx = -1; mx < 2; mx++) {
                if (my == 0 && mx == 0) continue;                                 /* :176 */
                int offsx = cx + mx * stepsize;
                if (offsx < 0 || offsx > refw - 16) continue;                     /* :182 */
                float err = calc_error(src, ref + (size_t)offsy * refw + offsx, refw, best_err);
                if (err < best_err) {                                             /* :189 strict */
                    best_err = err;
                    best_dx = mx * stepsize;
                    best_dy = my * stepsize;
                }
            }
        }
        cx += best_dx; cy += best_dy;                                             /* :199 */
        tot_dx += best_dx; tot_dy += best_dy;                                     /* :200 */
    }
    *out_dx = tot_dx; *out_dy = tot_dy; *out_err = best_err;
}

/* src/common.rs:206-236  encode_block_delta; returns has_coef */
static int encode_block_delta(const uint8_t src[256], const uint8_t *ref, int refw, int refh, int bx, int by,
                              const int32_t q[64], float px_err, int8_t mv[2], int16_t coef[256])
{
    float min_err = px_err * px_err * 256.0f;                                     /* :209 */
    int dx, dy; float best_err;
    block_search(src, ref, refw, refh, bx, by, &dx, &dy, &best_err);              /* :212 */
    mv[0] = (int8_t)dx; mv[1] = (int8_t)dy;
    if (best_err <= min_err) {                                                    /* :221 */
        memset(coef, 0, 256 * sizeof(int16_t));
        return 0;
    }
    /* src/common.rs:108-123 calc_residuals, clamp(-255,255) */
    int16_t delta[256];
    const uint8_t *prev = ref + (size_t)(by + dy) * refw + (bx + dx);
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++) {
            int d = (int)src[r * 16 + c] - (int)prev[r * refw + c];
            delta[r * 16 + c] = (int16_t)(d < -255 ? -255 : (d > 255 ? 255 : d));
        }
    encode_subblock_delta(delta + 0, 16, q, coef + 0);                            /* :228-232 */
    encode_subblock_delta(delta + 8, 16, q, coef + 64);
    encode_subblock_delta(delta + 8 * 16, 16, q, coef + 128);
    encode_subblock_delta(delta + 8 * 16 + 8, 16, q, coef + 192);
    return 1;
}

/* src/common.rs:254-285 + :327-339 + :98-104  decode_block_delta / get_block / apply_residuals */
static void decode_block_delta(const int8_t mv[2], int has_coef, const int16_t coef[256], const uint8_t *ref,
                               int refw, int bx, int by, const int32_t q[64], uint8_t mb[256])
{
    const uint8_t *prev = ref + (size_t)(by + mv[1]) * refw + (bx + mv[0]);
    if (has_coef) {
        decode_block(coef, q, mb);
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                int d = ((int)mb[r * 16 + c] - 128) * 2;                          /* :100 */
                int p = (int)prev[r * refw + c] + d;
                mb[r * 16 + c] = (uint8_t)(p < 0 ? 0 : (p > 255 ? 255 : p));     /* :102 */
            }
    } else {
        for (int r = 0; r < 16; r++) memcpy(mb + r * 16, prev + r * refw, 16);
    }
}

/* ---------------------------------------------------------------- fork/join helper
 * Stand-in for `tp.install(|| par_iter().map().collect())` (src/common.rs:374-378 and the
 * five sibling call sites): a persistent pool of `threads` workers (like the rayon pool the
 * reference builds once per Encoder/Decoder, src/enc.rs:54) that pull chunks of macroblock
 * indices from a shared counter; the caller blocks until the map is complete.  Results are
 * index-ordered, like rayon's collect(). */
typedef void (*mb_fn)(void *ctx, int mb_index);
#define PFVO_MAX_THREADS 512
#define PFVO_CHUNK 8
/* One fork/join.  It lives on the caller's stack; workers latch onto it under the pool mutex (users++) and the caller
 * leaves only when every index has been claimed (it drains the counter itself) and every latched worker has left
 * (users == 0).  A worker that wakes up after the job is over finds no current job and goes back to sleep: the join
 * never waits for threads that did no work -- like rayon, whose sleeping workers do not gate a finished scope.  (The
 * first version joined on ALL pool threads checking in; under the GPU box's CPU quota a 256-thread pool then spent its
 * time waking threads: 0.38 M macroblocks/s against 2.4 M with 32 threads.) */
typedef struct {
    mb_fn fn; void *ctx; int total;
    int next;               /* next unclaimed index (atomic) */
    int users;              /* workers inside this job (under the pool mutex) */
} pool_job;
static struct {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    pthread_t th[PFVO_MAX_THREADS];
    int n_threads;          /* workers alive */
    pool_job *cur;          /* job accepting workers, or NULL */
    int active;             /* workers allowed into the current job (indices < active) */
    int generation;         /* bumped per job */
    int quit;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, NULL, 0, 0, 0};

static void pool_run_chunks(pool_job *job)
{
    /* lock-free chunk claiming: the job was published under the mutex */
    for (;;) {
        int lo = __atomic_fetch_add(&job->next, PFVO_CHUNK, __ATOMIC_RELAXED);
        if (lo >= job->total) return;
        int hi = lo + PFVO_CHUNK > job->total ? job->total : lo + PFVO_CHUNK;
        for (int i = lo; i < hi; i++) job->fn(job->ctx, i);
    }
}
static void *pool_worker(void *arg)
{
    const int my_index = (int)(intptr_t)arg;
    int seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.generation == seen) pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
        seen = g_pool.generation;
        if (g_pool.quit) break;
        pool_job *job = g_pool.cur;
        if (!job || my_index >= g_pool.active) continue;   /* job already over, or it uses a smaller pool */
        job->users++;
        pthread_mutex_unlock(&g_pool.mu);
        pool_run_chunks(job);
        pthread_mutex_lock(&g_pool.mu);
        if (--job->users == 0) pthread_cond_broadcast(&g_pool.cv_done);
    }
    pthread_mutex_unlock(&g_pool.mu);
    return NULL;
}
/* joins every worker (so that a later, smaller pool is not slowed down by idle wake-ups) */
PFVO_API void pfvo_pool_shutdown(void)
{
    pthread_mutex_lock(&g_pool.mu);
    int n = g_pool.n_threads;
    g_pool.quit = 1;
    g_pool.generation++;
    pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.mu);
    for (int i = 0; i < n; i++) pthread_join(g_pool.th[i], NULL);
    pthread_mutex_lock(&g_pool.mu);
    g_pool.n_threads = 0;
    g_pool.quit = 0;
    pthread_mutex_unlock(&g_pool.mu);
}
static void par_for(int n, int threads, mb_fn fn, void *ctx)
{
    if (threads <= 1 || n < 2 * PFVO_CHUNK) {
        for (int i = 0; i < n; i++) fn(ctx, i);
        return;
    }
    if (threads > PFVO_MAX_THREADS) threads = PFVO_MAX_THREADS;
    pool_job job = {fn, ctx, n, 0, 0};
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.n_threads < threads - 1) {   /* the caller is the last "thread" */
        pthread_create(&g_pool.th[g_pool.n_threads], NULL, pool_worker, (void *)(intptr_t)g_pool.n_threads);
        g_pool.n_threads++;
    }
    g_pool.cur = &job;
    g_pool.active = threads - 1;
    g_pool.generation++;
    pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.mu);
    pool_run_chunks(&job);                     /* returns once every index is claimed */
    pthread_mutex_lock(&g_pool.mu);
    g_pool.cur = NULL;                         /* late wakers find nothing to do */
    while (job.users > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

static inline int pad16(int x) { return x + (16 - (x % 16)) % 16; } /* common.rs:352-353 */
PFVO_API int pfvo_pad16(int x) { return pad16(x); }

/* ---------------------------------------------------------------- plane level */
typedef struct {
    const uint8_t *blocks; /* gathered 16x16 macroblocks, 256 B each */
    const int32_t *q;
    int16_t *coef;
    /* delta only */
    const uint8_t *ref; int refw, refh, bw; float px_err; int8_t *mv; uint8_t *has_coef;
    /* decode only */
    uint8_t *out_blocks; const int16_t *in_coef; const int8_t *in_mv; const uint8_t *in_has;
} plane_ctx;

/* common.rs:352-370 / :389-407: pad + fill + blit, then gather one 16x16 block per MB */
static uint8_t *pad_and_gather(const uint8_t *px, int w, int h, uint8_t clear, int *pbw, int *pbh)
{
    int pw = pad16(w), ph = pad16(h), bw = pw / 16, bh = ph / 16;
    uint8_t *img = (uint8_t *)malloc((size_t)pw * ph);
    memset(img, clear, (size_t)pw * ph);
    for (int r = 0; r < h; r++) memcpy(img + (size_t)r * pw, px + (size_t)r * w, (size_t)w);
    uint8_t *blocks = (uint8_t *)malloc((size_t)bw * bh * 256);
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            uint8_t *b = blocks + ((size_t)by * bw + bx) * 256;
            for (int r = 0; r < 16; r++) memcpy(b + r * 16, img + (size_t)(by * 16 + r) * pw + bx * 16, 16);
        }
    free(img);
    *pbw = bw; *pbh = bh;
    return blocks;
}

static void job_encode_block(void *c, int i)
{
    plane_ctx *p = (plane_ctx *)c;
    encode_block(p->blocks + (size_t)i * 256, p->q, p->coef + (size_t)i * 256);
}

/* src/common.rs:351-386  VideoPlane::encode_plane
 * coef_out: [bw*bh][4][64] i16, zigzag order inside each subblock. */
PFVO_API void pfvo_encode_plane(const uint8_t *px, int w, int h, const int32_t q[64], uint8_t clear,
                                int16_t *coef_out, int threads)
{
    plane_ctx c; memset(&c, 0, sizeof c);
    int bw, bh;
    uint8_t *blocks = pad_and_gather(px, w, h, clear, &bw, &bh);
    c.blocks = blocks; c.q = q; c.coef = coef_out;
    par_for(bw * bh, threads, job_encode_block, &c);
    free(blocks);
}

static void job_encode_block_delta(void *c, int i)
{
    plane_ctx *p = (plane_ctx *)c;
    int bx = (i % p->bw) * 16, by = (i / p->bw) * 16;
    p->has_coef[i] = (uint8_t)encode_block_delta(p->blocks + (size_t)i * 256, p->ref, p->refw, p->refh, bx, by,
                                                 p->q, p->px_err, p->mv + (size_t)i * 2, p->coef + (size_t)i * 256);
}

/* src/common.rs:388-421  VideoPlane::encode_plane_delta
 * ref: the (padded) previous reconstructed plane, refw x refh. */
PFVO_API void pfvo_encode_plane_delta(const uint8_t *px, int w, int h, const uint8_t *ref, int refw, int refh,
                                      const int32_t q[64], float px_err, uint8_t clear, int8_t *mv_out,
                                      uint8_t *has_coef_out, int16_t *coef_out, int threads)
{
    plane_ctx c; memset(&c, 0, sizeof c);
    int bw, bh;
    uint8_t *blocks = pad_and_gather(px, w, h, clear, &bw, &bh);
    c.blocks = blocks; c.q = q; c.coef = coef_out; c.ref = ref; c.refw = refw; c.refh = refh; c.bw = bw;
    c.px_err = px_err; c.mv = mv_out; c.has_coef = has_coef_out;
    par_for(bw * bh, threads, job_encode_block_delta, &c);
    free(blocks);
}

static void job_decode_block(void *c, int i)
{
    plane_ctx *p = (plane_ctx *)c;
    decode_block(p->in_coef + (size_t)i * 256, p->q, p->out_blocks + (size_t)i * 256);
}

/* common.rs:438-443 etc.: serial scatter (blit_block :341-349) */
static void scatter(const uint8_t *blocks, int bw, int bh, uint8_t *plane)
{
    int pw = bw * 16;
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            const uint8_t *b = blocks + ((size_t)by * bw + bx) * 256;
            for (int r = 0; r < 16; r++) memcpy(plane + (size_t)(by * 16 + r) * pw + bx * 16, b + r * 16, 16);
        }
}

/* src/common.rs:423-446 decode_plane and :477-496 decode_plane_into (same result: every
 * pixel of the bw*16 x bh*16 target is overwritten) */
PFVO_API void pfvo_decode_plane_into(const int16_t *coef, int bw, int bh, const int32_t q[64], uint8_t *target,
                                     int threads)
{
    plane_ctx c; memset(&c, 0, sizeof c);
    uint8_t *blocks = (uint8_t *)malloc((size_t)bw * bh * 256);
    c.in_coef = coef; c.q = q; c.out_blocks = blocks;
    par_for(bw * bh, threads, job_decode_block, &c);
    scatter(blocks, bw, bh, target);
    free(blocks);
}

static void job_decode_block_delta(void *c, int i)
{
    plane_ctx *p = (plane_ctx *)c;
    int bx = (i % p->bw) * 16, by = (i / p->bw) * 16;                              /* :455-456 */
    decode_block_delta(p->in_mv + (size_t)i * 2, p->in_has[i], p->in_coef + (size_t)i * 256, p->ref, p->refw, bx, by,
                       p->q, p->out_blocks + (size_t)i * 256);
}

/* src/common.rs:448-475 decode_plane_delta (ref -> fresh out) and :498-521
 * decode_plane_delta_into (out == ref allowed: read-all-then-write-all).
 * Returns 0, or -1 if a motion vector points outside the reference plane (the reference
 * only debug_asserts this, common.rs:258-259; release builds would index out of range). */
PFVO_API int pfvo_decode_plane_delta(const int8_t *mv, const uint8_t *has_coef, const int16_t *coef, int bw, int bh,
                                     const int32_t q[64], const uint8_t *ref, uint8_t *out, int threads)
{
    int pw = bw * 16, ph = bh * 16;
    for (int i = 0; i < bw * bh; i++) {
        int sx = (i % bw) * 16 + mv[i * 2], sy = (i / bw) * 16 + mv[i * 2 + 1];
        if (sx < 0 || sx > pw - 16 || sy < 0 || sy > ph - 16) return -1;
    }
    plane_ctx c; memset(&c, 0, sizeof c);
    uint8_t *blocks = (uint8_t *)malloc((size_t)bw * bh * 256);
    c.in_coef = coef; c.in_mv = mv; c.in_has = has_coef; c.q = q; c.out_blocks = blocks;
    c.ref = ref; c.refw = pw; c.bw = bw;
    par_for(bw * bh, threads, job_decode_block_delta, &c);
    scatter(blocks, bw, bh, out); /* after the join: out may alias ref */
    free(blocks);
    return 0;
}

/* ---------------------------------------------------------------- VideoPlane::blit (plane.rs:20-29) */
PFVO_API void pfvo_blit(uint8_t *dst, int dstw, const uint8_t *src, int srcw, int dx, int dy, int sx, int sy, int sw,
                        int sh)
{
    for (int row = 0; row < sh; row++)
        memcpy(dst + (size_t)(row + dy) * dstw + dx, src + (size_t)(row + sy) * srcw + sx, (size_t)sw);
}

/* src/common.rs:523-536  VideoPlane::reduce: a (width / 2) x (height / 2) plane (integer division) holding the
 * pixel at (2 ix, 2 iy) of every 2 x 2 cell -- point sampling, no averaging. */
PFVO_API void pfvo_reduce(const uint8_t *src, int srcw, int srch, uint8_t *dst)
{
    int nw = srcw / 2, nh = srch / 2;
    for (int iy = 0; iy < nh; iy++)
        for (int ix = 0; ix < nw; ix++) {
            int sx = ix * 2, sy = iy * 2;
            dst[ix + (size_t)iy * nw] = src[sx + (size_t)sy * srcw];
        }
}

/* src/common.rs:538-556  VideoPlane::double: a (2 width) x (2 height) plane, every source pixel written to the four
 * positions d_idx, d_idx + 1, d_idx + new_width, d_idx + new_width + 1. */
PFVO_API void pfvo_double(const uint8_t *src, int srcw, int srch, uint8_t *dst)
{
    int nw = srcw * 2;
    for (int iy = 0; iy < srch; iy++)
        for (int ix = 0; ix < srcw; ix++) {
            int dx = ix * 2, dy = iy * 2;
            size_t d_idx = (size_t)dx + (size_t)dy * nw;
            uint8_t px = src[ix + (size_t)iy * srcw];
            dst[d_idx] = px;
            dst[d_idx + 1] = px;
            dst[d_idx + nw] = px;
            dst[d_idx + nw + 1] = px;
        }
}

/* ---------------------------------------------------------------- session level */
/* src/enc.rs:40-51  q-table derivation: max(1.0, base as f32 * qscale [* 0.5]) as i32 */
PFVO_API void pfvo_qtables(int quality, int32_t intra_l[64], int32_t intra_c[64], int32_t inter_l[64],
                           int32_t inter_c[64], float *px_err)
{
    float qscale = (float)quality * 0.25f;
    *px_err = (float)quality * 1.5f;
    for (int i = 0; i < 64; i++) {
        inter_l[i] = (int32_t)fmaxf((float)Q_TABLE_INTER_VALUE * qscale * 0.5f, 1.0f);
        inter_c[i] = (int32_t)fmaxf((float)Q_TABLE_INTER_VALUE * qscale, 1.0f);
        intra_l[i] = (int32_t)fmaxf((float)Q_TABLE_INTRA[i] * qscale * 0.5f, 1.0f);
        intra_c[i] = (int32_t)fmaxf((float)Q_TABLE_INTRA[i] * qscale, 1.0f);
    }
}

/* Encoder state restated from src/enc.rs:12-26 (hot-path fields only). */
typedef struct {
    int width, height, threads;
    int pw[3], ph[3];      /* padded dims per plane (frame.rs:28-49 new_padded) */
    uint8_t *prev[3];      /* prev_frame planes, padded */
    int32_t q_intra_l[64], q_intra_c[64], q_inter_l[64], q_inter_c[64];
    float px_err;
} pfvo_encoder;

PFVO_API pfvo_encoder *pfvo_encoder_new(int width, int height, int quality, int threads)
{
    if (quality < 0 || quality > 10) return NULL; /* enc.rs:38 assert */
    pfvo_encoder *e = (pfvo_encoder *)calloc(1, sizeof *e);
    e->width = width; e->height = height; e->threads = threads;
    e->pw[0] = pad16(width); e->ph[0] = pad16(height);
    e->pw[1] = e->pw[2] = pad16(width / 2); e->ph[1] = e->ph[2] = pad16(height / 2);
    for (int p = 0; p < 3; p++) {
        e->prev[p] = (uint8_t *)malloc((size_t)e->pw[p] * e->ph[p]);
        memset(e->prev[p], p == 0 ? 0 : 128, (size_t)e->pw[p] * e->ph[p]); /* frame.rs:38-43 */
    }
    pfvo_qtables(quality, e->q_intra_l, e->q_intra_c, e->q_inter_l, e->q_inter_c, &e->px_err);
    return e;
}
PFVO_API void pfvo_encoder_free(pfvo_encoder *e)
{
    if (!e) return;
    for (int p = 0; p < 3; p++) free(e->prev[p]);
    free(e);
}
PFVO_API int pfvo_encoder_total_blocks(const pfvo_encoder *e)
{
    int n = 0;
    for (int p = 0; p < 3; p++) n += (e->pw[p] / 16) * (e->ph[p] / 16);
    return n;
}
PFVO_API const uint8_t *pfvo_encoder_prev_plane(const pfvo_encoder *e, int p, int *pw, int *ph)
{
    *pw = e->pw[p]; *ph = e->ph[p];
    return e->prev[p];
}

/* src/enc.rs:75-123 encode_iframe, hot-path part (:84-97): per plane encode_plane ->
 * decode_plane -> prev_frame.blit.  coef_out: [total_blocks][256] in plane order Y,U,V. */
PFVO_API void pfvo_encode_iframe(pfvo_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                                 int16_t *coef_out)
{
    const uint8_t *src[3] = {y, u, v};
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        int w = p == 0 ? e->width : e->width / 2, h = p == 0 ? e->height : e->height / 2;
        const int32_t *q = p == 0 ? e->q_intra_l : e->q_intra_c;
        int bw = e->pw[p] / 16, bh = e->ph[p] / 16;
        pfvo_encode_plane(src[p], w, h, q, p == 0 ? 0 : 128, coef_out + off * 256, e->threads);
        pfvo_decode_plane_into(coef_out + off * 256, bw, bh, q, e->prev[p], e->threads);
        off += (size_t)bw * bh;
    }
}

/* src/enc.rs:125-173 encode_pframe, hot-path part (:134-147). */
PFVO_API void pfvo_encode_pframe(pfvo_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                                 int8_t *mv_out, uint8_t *has_coef_out, int16_t *coef_out)
{
    const uint8_t *src[3] = {y, u, v};
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        int w = p == 0 ? e->width : e->width / 2, h = p == 0 ? e->height : e->height / 2;
        const int32_t *q = p == 0 ? e->q_inter_l : e->q_inter_c;
        int bw = e->pw[p] / 16, bh = e->ph[p] / 16;
        pfvo_encode_plane_delta(src[p], w, h, e->prev[p], e->pw[p], e->ph[p], q, e->px_err, p == 0 ? 0 : 128,
                                mv_out + off * 2, has_coef_out + off, coef_out + off * 256, e->threads);
        pfvo_decode_plane_delta(mv_out + off * 2, has_coef_out + off, coef_out + off * 256, bw, bh, q, e->prev[p],
                                e->prev[p], e->threads);
        off += (size_t)bw * bh;
    }
}

/* Decoder state restated from src/dec.rs:15-28 (hot-path fields only): q-tables + padded framebuffer. */
typedef struct {
    int width, height, threads, n_qtables;
    int pw[3], ph[3];
    uint8_t *fb[3];
    int32_t *qtables; /* n_qtables x 64 */
} pfvo_decoder;

PFVO_API pfvo_decoder *pfvo_decoder_new(int width, int height, const int32_t *qtables, int n_qtables, int threads)
{
    pfvo_decoder *d = (pfvo_decoder *)calloc(1, sizeof *d);
    d->width = width; d->height = height; d->threads = threads; d->n_qtables = n_qtables;
    d->pw[0] = pad16(width); d->ph[0] = pad16(height);
    d->pw[1] = d->pw[2] = pad16(width / 2); d->ph[1] = d->ph[2] = pad16(height / 2);
    for (int p = 0; p < 3; p++) { /* VideoFrame::new_padded, src/dec.rs:123 + frame.rs:38-43 */
        d->fb[p] = (uint8_t *)malloc((size_t)d->pw[p] * d->ph[p]);
        memset(d->fb[p], p == 0 ? 0 : 128, (size_t)d->pw[p] * d->ph[p]);
    }
    d->qtables = (int32_t *)malloc((size_t)n_qtables * 64 * sizeof(int32_t));
    memcpy(d->qtables, qtables, (size_t)n_qtables * 64 * sizeof(int32_t));
    return d;
}
PFVO_API void pfvo_decoder_free(pfvo_decoder *d)
{
    if (!d) return;
    for (int p = 0; p < 3; p++) free(d->fb[p]);
    free(d->qtables);
    free(d);
}
PFVO_API const uint8_t *pfvo_decoder_plane(const pfvo_decoder *d, int p, int *pw, int *ph)
{
    *pw = d->pw[p]; *ph = d->ph[p];
    return d->fb[p];
}
/* src/dec.rs:298-323 (after entropy decoding): deserialize_plane x3 -> decode_plane_into */
PFVO_API void pfvo_decode_iframe(pfvo_decoder *d, const int16_t *coef, const uint8_t qidx[3])
{
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        int bw = d->pw[p] / 16, bh = d->ph[p] / 16;
        pfvo_decode_plane_into(coef + off * 256, bw, bh, d->qtables + (size_t)qidx[p] * 64, d->fb[p], d->threads);
        off += (size_t)bw * bh;
    }
}
/* src/dec.rs:419-445: deserialize_plane_delta x3 -> decode_plane_delta_into */
PFVO_API int pfvo_decode_pframe(pfvo_decoder *d, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef,
                                const uint8_t qidx[3])
{
    size_t off = 0;
    for (int p = 0; p < 3; p++) {
        int bw = d->pw[p] / 16, bh = d->ph[p] / 16;
        int rc = pfvo_decode_plane_delta(mv + off * 2, has_coef + off, coef + off * 256, bw, bh,
                                         d->qtables + (size_t)qidx[p] * 64, d->fb[p], d->fb[p], d->threads);
        if (rc) return rc;
        off += (size_t)bw * bh;
    }
    return 0;
}


/* ------------------------------------------------------------------ colour helpers of the reference's tests
 * load_frame (src/lib.rs:337-359) + VideoFrame::from_planes (src/frame.rs:51-59), save_frame (src/lib.rs:361-394).
 * f32 arithmetic in source order; `as u8` = truncate toward zero, saturating, NaN -> 0. */
static uint8_t f32_as_u8(float x) { return x >= 255.0f ? 255 : (x > 0.0f ? (uint8_t)(int)x : 0); }
PFVO_API void pfvo_rgb_to_yuv420(const uint8_t *rgb, int w, int h, uint8_t *frame)
{
    size_t n = (size_t)w * h;
    int cw = w / 2, ch = h / 2;
    uint8_t *py = frame, *pu = frame + n, *pv = pu + (size_t)cw * ch;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            volatile float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
            volatile float t0 = 0.299f * r, t1 = 0.587f * g, t2 = 0.114f * b;
            volatile float yy = t0 + t1;
            yy = yy + t2;
            py[i] = f32_as_u8(yy);
            if (!(x & 1) && !(y & 1) && x / 2 < cw && y / 2 < ch) {          /* reduce(): pixel (2x, 2y) (src/common.rs:523-536) */
                volatile float u0 = 0.168736f * r, u1 = 0.331264f * g, u2 = 0.5f * b;
                volatile float u = 128.0f - u0;
                u = u - u1;
                u = u + u2;
                volatile float v0 = 0.5f * r, v1 = 0.418688f * g, v2 = 0.081312f * b;
                volatile float v = 128.0f + v0;
                v = v - v1;
                v = v - v2;
                pu[(size_t)(y / 2) * cw + x / 2] = f32_as_u8(u);
                pv[(size_t)(y / 2) * cw + x / 2] = f32_as_u8(v);
            }
        }
}
PFVO_API void pfvo_yuv420_to_rgb(const uint8_t *frame, int w, int h, uint8_t *rgb)
{
    size_t n = (size_t)w * h;
    int cw = w / 2, ch = h / 2;
    const uint8_t *py = frame, *pu = frame + n, *pv = pu + (size_t)cw * ch;
    (void)ch;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x, c = (size_t)(y / 2) * cw + x / 2;      /* double(): nearest (src/common.rs:538-556) */
            volatile float yy = py[i], u = (float)pu[c] - 128.0f, v = (float)pv[c] - 128.0f;
            volatile float r1 = 1.402f * v, g1 = 0.344136f * u, g2 = 0.714136f * v, b1 = 1.772f * u;
            volatile float r = yy + r1, g = yy - g1, b = yy + b1;
            g = g - g2;
            rgb[3 * i] = f32_as_u8(r);
            rgb[3 * i + 1] = f32_as_u8(g);
            rgb[3 * i + 2] = f32_as_u8(b);
        }
}
