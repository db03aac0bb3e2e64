// pfv_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels for the Pretty Fast Video
// per-macroblock transform / motion path.  No MFMA: the transforms are 36-add / 12-shift
// integer butterflies with per-stage truncation (reference src/dct.rs:176-293), and the
// motion search is a data-dependent 4-step descent (src/common.rs:154-204).
//
// Work decomposition (all kernels):
//   workgroup  = 256 threads = 4 wavefronts = one STRIP of 8 horizontally adjacent
//                macroblocks (128 x 16 pixels) of one plane of one stream;
//   wavefront  = 2 macroblocks, 32 lanes each  (h = lane >> 5);
//   transform  : lane (s, r) = (l >> 3, l & 7) owns row r of 8x8 subblock s in "row
//                layout" and column r of it in "column layout"; the 1-D butterflies run
//                entirely in registers, the 8x8 transposes between the two passes go
//                through a wavefront-private LDS scratch (conflict-free column reads);
//   search     : lane (cand, j) = (l >> 2, l & 3): 8 candidates of one search level are
//                evaluated at once, 4 lanes x 4 rows each; the 16x16 source block lives in
//                registers, the +-15 pixel reference window lives in LDS; squared error is
//                built from v_dot4_u32_u8 as  sum a^2 - 2 sum ab + sum b^2  (exact in u32);
//   memory     : every HBM access is a 16-byte-per-lane coalesced vector load/store of a
//                128-byte plane row segment or of the strip's contiguous 4 KiB coefficient
//                block; unaligned 16x16 reference patches are cut out of LDS with
//                v_alignbyte_b32.
//
// Bit-exactness notes (reference line numbers in the function comments):
//   - i32 arithmetic wraps (Rust release) -> done in unsigned here;
//   - `/` truncates toward zero, `>>` is arithmetic;
//   - encode = rows then columns, decode = columns then rows;
//   - encode indexes SCALE/q by raster index, decode by zigzag position (QTab::deq).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfv_device.h"

namespace pfv {

// reference src/dct.rs:4-13 (data)
__constant__ int kScale[64] = {
    32, 37, 34, 26, 32, 26, 34, 37, 37, 43, 39, 31, 37, 31, 39, 43, 34, 39, 35, 28, 34, 28, 35, 39, 26, 31, 28, 22, 26, 22,
    28, 31, 32, 37, 34, 26, 32, 26, 34, 37, 26, 31, 28, 22, 26, 22, 28, 31, 34, 39, 35, 28, 34, 28, 35, 39, 37, 43, 39, 31,
    37, 31, 39, 43,
};
// reference src/dct.rs:39-42 (data): raster index -> zigzag position
__constant__ int kInvZigzag[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40,
    44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49,
    57, 58, 62, 63,
};

// ------------------------------------------------------------------ LDS layout
constexpr int kRefRows = 46;            // 16 + 2*15 rows of reference window
constexpr int kRefStride = 176;         // bytes per window row: 160 used (x0-16 .. x0+143), 16-byte aligned
constexpr int kRefBytes = kRefRows * kRefStride;
constexpr int kTileBytes = 16 * 128;    // source strip, later overwritten by the reconstructed strip
constexpr int kCoefBytes = kStripMB * 512;
constexpr int kSubStride = 72;          // dwords per subblock in the transpose scratch (64 + 8 pad)
constexpr int kScratchDwords = 8 * kSubStride;   // per wavefront: 2 MB x 4 subblocks

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
// Rust `/` by 2, 4, 16 on i32: truncation toward zero
__device__ __forceinline__ int tdiv2(int x) { return (int)((unsigned)x + ((unsigned)x >> 31)) >> 1; }
__device__ __forceinline__ int tdiv4(int x) { return (int)((unsigned)x + ((unsigned)(x >> 31) >> 30)) >> 2; }
__device__ __forceinline__ int tdiv16(int x) { return (int)((unsigned)x + ((unsigned)(x >> 31) >> 28)) >> 4; }

// Intra-wavefront LDS hand-off: DS operations of one wavefront execute in issue order, so
// only the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP cross-lane moves (bound_ctrl: out-of-range source lanes read 0; never happens for these)
template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kRowHalfMirror = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int kRowMirror = 0x140;     // lane i <-> 15-i inside each 16 lanes

__device__ __forceinline__ int quad_sum(int v)
{
    v += dpp<kQuadXor1>(v);
    v += dpp<kQuadXor2>(v);
    return v;
}
// min over the 32 lanes of a half-wavefront; input must already be uniform inside each quad
__device__ __forceinline__ unsigned half_min_of_quads(unsigned v)
{
    v = min(v, (unsigned)dpp<kRowHalfMirror>((int)v));
    v = min(v, (unsigned)dpp<kRowMirror>((int)v));
    v = min(v, (unsigned)__shfl_xor((int)v, 16));
    return v;
}
// sum over the 32 lanes of a half-wavefront
__device__ __forceinline__ int half_sum(int v)
{
    v = quad_sum(v);
    v += dpp<kRowHalfMirror>(v);
    v += dpp<kRowMirror>(v);
    v += __shfl_xor(v, 16);
    return v;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (observed, speed only).
// Give every XCD one contiguous range of strips so that vertically / horizontally adjacent
// strips -- which share reference-window halos -- meet in the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    int xcd = b & 7, idx = b >> 3;
    int per = nb >> 3, rem = nb & 7;
    return xcd * per + min(xcd, rem) + idx;
}

struct StripPos {
    int stream, plane;
    int sx, by;    // strip column index, macroblock row
    int x0, y0;    // pixel origin of the strip in the padded plane
    int n_mb;      // macroblocks of this strip that exist (1..8)
    int mb_first;  // frame-relative index of the strip's first macroblock
};

__device__ __forceinline__ StripPos locate_strip(const FrameGeom &g)
{
    int vb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    StripPos s;
    s.stream = vb / g.strips_per_frame;
    int st = vb - s.stream * g.strips_per_frame;
    s.plane = (g.n_planes > 2 && st >= g.p[2].strip0) ? 2 : ((g.n_planes > 1 && st >= g.p[1].strip0) ? 1 : 0);
    const PlaneGeom &p = g.p[s.plane];
    st -= p.strip0;
    s.by = st / p.strips_x;
    s.sx = st - s.by * p.strips_x;
    s.x0 = s.sx * (kStripMB * 16);
    s.y0 = s.by * 16;
    s.n_mb = min(kStripMB, p.bw - s.sx * kStripMB);
    s.mb_first = p.mb0 + s.by * p.bw + s.sx * kStripMB;
    return s;
}

// ------------------------------------------------------------------ 1-D transforms
// reference src/dct.rs:176-239  DctMatrix8x8::fdct
__device__ __forceinline__ void fdct8(int (&v)[8])
{
    int a0 = wadd(v[0], v[7]), a1 = wadd(v[1], v[6]), a2 = wadd(v[2], v[5]), a3 = wadd(v[3], v[4]);
    int a4 = wsub(v[0], v[7]), a5 = wsub(v[1], v[6]), a6 = wsub(v[2], v[5]), a7 = wsub(v[3], v[4]);
    int b0 = wadd(a0, a3), b1 = wadd(a1, a2), b2 = wsub(a0, a3), b3 = wsub(a1, a2);
    int c0 = wadd(b0, b1), c1 = wsub(b0, b1);
    int c2 = wadd(wadd(b2, tdiv4(b2)), tdiv2(b3));
    int c3 = wsub(wsub(tdiv2(b2), b3), tdiv4(b3));
    int a4q = tdiv4(a4), a7q = tdiv4(a7);
    int b4 = wsub(wadd(wadd(a7q, a4), a4q), tdiv16(a4));
    int b7 = wadd(wsub(wsub(a4q, a7), a7q), tdiv16(a7));
    int b5 = wsub(wsub(wadd(a5, a6), tdiv4(a6)), tdiv16(a6));
    int b6 = wadd(wadd(wsub(a6, a5), tdiv4(a5)), tdiv16(a5));
    int c4 = wadd(b4, b5), c5 = wsub(b4, b5), c6 = wadd(b6, b7), c7 = wsub(b6, b7);
    v[0] = c0; v[1] = c4; v[2] = c2; v[3] = wsub(c5, c7);
    v[4] = c1; v[5] = wadd(c5, c7); v[6] = c3; v[7] = c6;
}

// reference src/dct.rs:241-293  DctMatrix8x8::idct
__device__ __forceinline__ void idct8(int (&v)[8])
{
    int c0 = v[0], d4 = v[1], c2 = v[2], d6 = v[3], c1 = v[4], d5 = v[5], c3 = v[6], d7 = v[7];
    int c4 = d4, c5 = wadd(d5, d6), c7 = wsub(d5, d6), c6 = d7;
    int b4 = wadd(c4, c5), b5 = wsub(c4, c5), b6 = wadd(c6, c7), b7 = wsub(c6, c7);
    int b0 = wadd(c0, c1), b1 = wsub(c0, c1);
    int b2 = wadd(wadd(c2, tdiv4(c2)), tdiv2(c3));
    int b3 = wsub(wsub(tdiv2(c2), c3), tdiv4(c3));
    int b4q = tdiv4(b4), b7q = tdiv4(b7);
    int a4 = wsub(wadd(wadd(b7q, b4), b4q), tdiv16(b4));
    int a7 = wadd(wsub(wsub(b4q, b7), b7q), tdiv16(b7));
    int a5 = wadd(wadd(wsub(b5, b6), tdiv4(b6)), tdiv16(b6));
    int a6 = wsub(wsub(wadd(b6, b5), tdiv4(b5)), tdiv16(b5));
    int a0 = wadd(b0, b2), a1 = wadd(b1, b3), a2 = wsub(b1, b3), a3 = wsub(b0, b2);
    v[0] = wadd(a0, a4); v[1] = wadd(a1, a5); v[2] = wadd(a2, a6); v[3] = wadd(a3, a7);
    v[4] = wsub(a3, a7); v[5] = wsub(a2, a6); v[6] = wsub(a1, a5); v[7] = wsub(a0, a4);
}

// ------------------------------------------------------------------ 8x8 transposes through LDS
// scratch layout per wavefront: T[h][s][row][col] as dwords, subblock stride kSubStride.
// lane (h, s, i): `sub` points at T[h][s][0][0].
// row layout (lane holds M[i][0..7])  ->  column layout (lane holds M[0..7][i])
__device__ __forceinline__ void rows_to_cols(int (&v)[8], int *sub, int i)
{
    int4 *w = reinterpret_cast<int4 *>(sub + i * 8);
    w[0] = make_int4(v[0], v[1], v[2], v[3]);
    w[1] = make_int4(v[4], v[5], v[6], v[7]);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = sub[k * 8 + i];
    wave_lds_sync();
}
// column layout (lane holds M[0..7][i])  ->  row layout (lane holds M[i][0..7])
__device__ __forceinline__ void cols_to_rows(int (&v)[8], int *sub, int i)
{
#pragma unroll
    for (int k = 0; k < 8; k++) sub[k * 8 + i] = v[k];
    wave_lds_sync();
    const int4 *w = reinterpret_cast<const int4 *>(sub + i * 8);
    int4 lo = w[0], hi = w[1];
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    wave_lds_sync();
}

// Per-lane quantiser constants for column c of a subblock (raster indices k*8 + c).
struct LaneQ {
    int scale[8];   // DCT_SCALE_FACTOR[k*8+c]                    (encode, raster-indexed)
    float rcp[8];   // biased 1/q[k*8+c]                          (encode, raster-indexed)
    int deq[8];     // SCALE[z]*q[z], z = INV_ZIGZAG[k*8+c]        (decode, zigzag-position-indexed)
    int zz[8];      // INV_ZIGZAG[k*8+c]: where coefficient (k,c) sits in the 64-entry zigzag run
};
template <bool ENC>
__device__ __forceinline__ void load_lane_q(LaneQ &lq, const QTab *qt, int c)
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int idx = k * 8 + c;
        lq.zz[k] = kInvZigzag[idx];
        lq.deq[k] = qt->deq[idx];
        if (ENC) {
            lq.scale[k] = kScale[idx];
            lq.rcp[k] = qt->rcp[idx];
        }
    }
}

// reference src/dct.rs:88-99  DctMatrix8x8::encode, for the lane's column:
//   n = (m * SCALE) >> 16 (arithmetic), out = n / q (truncating), stored as i16.
// The division is float(n) * rcp followed by a truncating convert; exact for |n| <= 2^15
// (QTab::rcp).  |n| <= 5160 for any u8 input (|m| <= 240 * 32768, SCALE <= 43).
__device__ __forceinline__ void quantize_col(const int (&v)[8], const LaneQ &lq, int (&qc)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int n = wmul(v[k], lq.scale[k]) >> 16;
        int q = (int)((float)n * lq.rcp[k]);
        qc[k] = (int)(short)q;
    }
}
// reference src/dct.rs:75-86  DctMatrix8x8::decode, for the lane's column
__device__ __forceinline__ void dequantize_col(const int (&qc)[8], const LaneQ &lq, int (&v)[8])
{
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = wmul(qc[k], lq.deq[k]);
}

// reference src/common.rs:313-325 decode_subblock tail: ((v >> 8) + 128).clamp(0,255)
__device__ __forceinline__ int to_pixel(int v) { return min(max(wadd(v >> 8, 128), 0), 255); }

__device__ __forceinline__ uint2 pack8(const int (&p)[8])
{
    uint2 o;
    o.x = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
    o.y = (unsigned)p[4] | ((unsigned)p[5] << 8) | ((unsigned)p[6] << 16) | ((unsigned)p[7] << 24);
    return o;
}
__device__ __forceinline__ int byte_of(unsigned w, int i) { return (int)((w >> (8 * i)) & 0xffu); }

// ------------------------------------------------------------------ cooperative strip I/O
// Load the 128 x 16 source strip into LDS, applying the reference's pad rule
// (src/common.rs:352-356: img_copy.fill(clear); blit(source)).
__device__ __forceinline__ void load_src_strip(uint8_t *tile, const uint8_t *plane, const PlaneGeom &p, int x0, int y0)
{
    int t = threadIdx.x;
    if (t < 128) {
        int row = t >> 3, ch = t & 7;
        int x = x0 + ch * 16, y = y0 + row;
        uint4 val;
        unsigned fill = (unsigned)p.clear * 0x01010101u;
        val = make_uint4(fill, fill, fill, fill);
        if (y < p.h && x < p.w) {
            const uint8_t *src = plane + (long)y * p.w + x;
            if (p.fast_src && x + 16 <= p.w) {
                val = *reinterpret_cast<const uint4 *>(src);
            } else {
                unsigned wds[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    unsigned acc = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        int xx = x + i * 4 + b;
                        unsigned px = xx < p.w ? (unsigned)src[i * 4 + b] : (unsigned)p.clear;
                        acc |= px << (8 * b);
                    }
                    wds[i] = acc;
                }
                val = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            }
        }
        *reinterpret_cast<uint4 *>(tile + row * 128 + ch * 16) = val;
    }
}

// Store the reconstructed strip (LDS tile) to the padded plane: 128 lanes x 16 B.
__device__ __forceinline__ void store_recon_strip(const uint8_t *tile, uint8_t *plane, const PlaneGeom &p, int x0, int y0,
                                                  int n_mb)
{
    int t = threadIdx.x;
    if (t < 128) {
        int row = t >> 3, ch = t & 7;
        if (ch < n_mb) {
            uint4 val = *reinterpret_cast<const uint4 *>(tile + row * 128 + ch * 16);
            *reinterpret_cast<uint4 *>(plane + (long)(y0 + row) * p.pw + x0 + ch * 16) = val;
        }
    }
}

// Store / load the strip's coefficient block: n_mb * 512 contiguous bytes, 16 B per lane.
__device__ __forceinline__ void store_coef_strip(const uint8_t *stage, int16_t *coef_mb0, int n_mb)
{
    int t = threadIdx.x;
    if ((t >> 5) < n_mb) reinterpret_cast<uint4 *>(coef_mb0)[t] = reinterpret_cast<const uint4 *>(stage)[t];
}
__device__ __forceinline__ void load_coef_strip(uint8_t *stage, const int16_t *coef_mb0, int n_mb)
{
    int t = threadIdx.x;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((t >> 5) < n_mb) v = reinterpret_cast<const uint4 *>(coef_mb0)[t];
    reinterpret_cast<uint4 *>(stage)[t] = v;
}

// ------------------------------------------------------------------ subblock pipelines (per lane)
// Forward: 8 row-layout inputs (24.8 fixed point) -> quantised column (qc) + zigzag scatter
// into the strip's coefficient stage.  reference src/common.rs:294-297.
__device__ __forceinline__ void forward_subblock(int (&v)[8], int *sub, int i, const LaneQ &lq, int (&qc)[8],
                                                 int16_t *stage_sub)
{
    fdct8(v);                 // dct_transform_rows
    rows_to_cols(v, sub, i);
    fdct8(v);                 // dct_transform_columns
    quantize_col(v, lq, qc);
#pragma unroll
    for (int k = 0; k < 8; k++) stage_sub[lq.zz[k]] = (int16_t)qc[k];
}
// Inverse: quantised column -> 8 row-layout "pixel" values in 0..255.
// reference src/common.rs:314-322 (columns first, then rows).
__device__ __forceinline__ void inverse_subblock(const int (&qc)[8], int *sub, int i, const LaneQ &lq, int (&px)[8])
{
    int v[8];
    dequantize_col(qc, lq, v);
    idct8(v);                 // dct_inverse_transform_columns
    cols_to_rows(v, sub, i);
    idct8(v);                 // dct_inverse_transform_rows
#pragma unroll
    for (int k = 0; k < 8; k++) px[k] = to_pixel(v[k]);
}

// ================================================================== I-frame encode (+ closed-loop reconstruction)
// reference: VideoPlane::encode_plane (src/common.rs:351-386) fused with the
// VideoPlane::decode_plane that Encoder::encode_iframe runs on its output
// (src/enc.rs:84-97).  recon == nullptr -> encode only (plane-level operator).
__global__ __launch_bounds__(kThreads) void k_enc_iframe(FrameGeom g, const uint8_t *__restrict__ src,
                                                          int16_t *__restrict__ coef, uint8_t *__restrict__ recon,
                                                          const QTab *__restrict__ qtabs)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kCoefBytes];
    __shared__ __attribute__((aligned(16))) int scratch[4 * kScratchDwords];

    const StripPos sp = locate_strip(g);
    const PlaneGeom &p = g.p[sp.plane];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31, s = l >> 3, i = l & 7;
    const int mbl = wave * 2 + h;   // macroblock inside the strip

    LaneQ lq;
    load_lane_q<true>(lq, qtabs + p.qsel, i);

    load_src_strip(tile, src + (long)sp.stream * g.src_frame_bytes + p.src_off, p, sp.x0, sp.y0);
    __syncthreads();

    // lane's 8 source pixels: row (s>>1)*8 + i, columns (s&1)*8 .. +7 of macroblock mbl
    const int prow = (s >> 1) * 8 + i, pcol = mbl * 16 + (s & 1) * 8;
    uint2 raw = *reinterpret_cast<const uint2 *>(tile + prow * 128 + pcol);
    int v[8], qc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int px = byte_of(k < 4 ? raw.x : raw.y, k & 3);
        v[k] = (int)((unsigned)(px - 128) << 8);    // src/common.rs:291
    }
    int *sub = scratch + wave * kScratchDwords + (h * 4 + s) * kSubStride;
    int16_t *stage_sub = reinterpret_cast<int16_t *>(stage) + mbl * 256 + s * 64;
    forward_subblock(v, sub, i, lq, qc, stage_sub);

    if (recon) {
        int px[8];
        inverse_subblock(qc, sub, i, lq, px);
        *reinterpret_cast<uint2 *>(tile + prow * 128 + pcol) = pack8(px);
    }
    __syncthreads();

    store_coef_strip(stage, coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256, sp.n_mb);
    if (recon) store_recon_strip(tile, recon + (long)sp.stream * g.pad_frame_bytes + p.pad_off, p, sp.x0, sp.y0, sp.n_mb);
}

// ================================================================== P-frame encode (+ closed-loop reconstruction)
// reference: VideoPlane::encode_plane_delta (src/common.rs:388-421) -> encode_block_delta
// (:206-236) -> block_search (:154-204) / calc_error (:125-139) / calc_residuals (:108-123)
// / encode_subblock_delta (:300-311), fused with the decode_plane_delta (:448-475,
// decode_block_delta :254-285, apply_residuals :98-104) that Encoder::encode_pframe runs on
// the result (src/enc.rs:134-147).  recon == nullptr -> encode only.
__global__ __launch_bounds__(kThreads) void k_enc_pframe(FrameGeom g, const uint8_t *__restrict__ src,
                                                          const uint8_t *__restrict__ ref, int8_t *__restrict__ mv_out,
                                                          uint8_t *__restrict__ has_out, int16_t *__restrict__ coef,
                                                          uint8_t *__restrict__ recon, const QTab *__restrict__ qtabs,
                                                          float min_err)
{
    __shared__ __attribute__((aligned(16))) uint8_t win[kRefBytes];
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kCoefBytes];
    __shared__ __attribute__((aligned(16))) int scratch[4 * kScratchDwords];

    const StripPos sp = locate_strip(g);
    const PlaneGeom &p = g.p[sp.plane];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31;
    const int mbl = wave * 2 + h;
    const bool mb_valid = mbl < sp.n_mb;

    const uint8_t *refp = ref + (long)sp.stream * g.pad_frame_bytes + p.pad_off;
    const int winx0 = sp.x0 - 16, winy0 = sp.y0 - 15;

    // ---- stage the source strip and the reference window
    load_src_strip(tile, src + (long)sp.stream * g.src_frame_bytes + p.src_off, p, sp.x0, sp.y0);
    for (int c = threadIdx.x; c < kRefRows * 10; c += kThreads) {
        int row = c / 10, ch = c - row * 10;
        int y = winy0 + row, x = winx0 + ch * 16;
        if (y >= 0 && y < p.ph && x >= 0 && x < p.pw) {
            uint4 val = *reinterpret_cast<const uint4 *>(refp + (long)y * p.pw + x);
            *reinterpret_cast<uint4 *>(win + row * kRefStride + ch * 16) = val;
        }
    }
    __syncthreads();

    // ---- motion search: lane (cand, j); macroblock origin in plane / window coordinates
    const int cand = l >> 2, j = l & 3;
    const int k9 = cand < 4 ? cand : cand + 1;      // skip the centre slot of the 3x3 pattern
    const int cmy = k9 / 3 - 1, cmx = k9 - (k9 / 3) * 3 - 1;   // visiting order: my outer, mx inner (:168-175)
    const int mbx = sp.x0 + mbl * 16, mby = sp.y0;

    // source rows j, j+4, j+8, j+12 in registers
    uint4 a[4];
    int a2 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        a[k] = *reinterpret_cast<const uint4 *>(tile + (j + 4 * k) * 128 + mbl * 16);
        a2 = (int)__builtin_amdgcn_udot4(a[k].x, a[k].x, (unsigned)a2, false);
        a2 = (int)__builtin_amdgcn_udot4(a[k].y, a[k].y, (unsigned)a2, false);
        a2 = (int)__builtin_amdgcn_udot4(a[k].z, a[k].z, (unsigned)a2, false);
        a2 = (int)__builtin_amdgcn_udot4(a[k].w, a[k].w, (unsigned)a2, false);
    }
    a2 = quad_sum(a2);   // sum of squares of the whole source block

    // evaluates sum b^2 - 2 sum ab over the lane's 4 rows for the patch at window coords (wx, wy)
    auto partial_err = [&](int wx, int wy) -> int {
        const int sh = wx & 3;
        const uint8_t *base = win + (wy + j) * kRefStride + (wx & ~3);
        unsigned ab = 0, bb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned *d = reinterpret_cast<const unsigned *>(base + 4 * k * kRefStride);
            unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
            unsigned b0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
            unsigned b1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            unsigned b2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            unsigned b3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            ab = __builtin_amdgcn_udot4(a[k].x, b0, ab, false);
            ab = __builtin_amdgcn_udot4(a[k].y, b1, ab, false);
            ab = __builtin_amdgcn_udot4(a[k].z, b2, ab, false);
            ab = __builtin_amdgcn_udot4(a[k].w, b3, ab, false);
            bb = __builtin_amdgcn_udot4(b0, b0, bb, false);
            bb = __builtin_amdgcn_udot4(b1, b1, bb, false);
            bb = __builtin_amdgcn_udot4(b2, b2, bb, false);
            bb = __builtin_amdgcn_udot4(b3, b3, bb, false);
        }
        return (int)bb - 2 * (int)ab;
    };

    // centre of the first level (src/common.rs:161-165): every quad computes it (cheap, once)
    int cx = 0, cy = 0;   // accumulated displacement
    int cur_err = a2 + quad_sum(partial_err(mbx - winx0, mby - winy0));

#pragma unroll 1
    for (int step = 8; step >= 1; step >>= 1) {
        int ox = mbx + cx + cmx * step, oy = mby + cy + cmy * step;
        bool valid = ox >= 0 && ox <= p.pw - 16 && oy >= 0 && oy <= p.ph - 16;   // :171, :182
        int ex = valid ? ox : mbx + cx, ey = valid ? oy : mby + cy;
        int err = a2 + quad_sum(partial_err(ex - winx0, ey - winy0));
        // strict `<` with first-visited-wins (:189)  ==  lexicographic min of (err, visiting order),
        // the centre being order 0
        unsigned key = valid ? (((unsigned)err << 4) | (unsigned)(cand + 1)) : 0xffffffffu;
        key = half_min_of_quads(key);
        key = min(key, (unsigned)cur_err << 4);
        int ord = (int)(key & 15u);
        cur_err = (int)(key >> 4);
        if (ord) {
            int b9 = ord - 1;
            b9 = b9 < 4 ? b9 : b9 + 1;
            cy += (b9 / 3 - 1) * step;
            cx += (b9 - (b9 / 3) * 3 - 1) * step;
        }
    }

    // ---- skip decision (src/common.rs:209, :221): best_err <= px_err^2 * 256, compared in f32
    const bool coded = mb_valid && !((float)cur_err <= min_err);
    if (l == 0 && mb_valid) {
        long mbi = (long)sp.stream * g.mbs_per_frame + sp.mb_first + mbl;
        mv_out[mbi * 2 + 0] = (int8_t)cx;
        mv_out[mbi * 2 + 1] = (int8_t)cy;
        has_out[mbi] = coded ? 1 : 0;
    }

    // ---- residual transform + reconstruction; lane (s, i)
    const int s = l >> 3, i = l & 7;
    const int prow = (s >> 1) * 8 + i, pcol = (s & 1) * 8;
    int16_t *stage_sub = reinterpret_cast<int16_t *>(stage) + mbl * 256 + s * 64;

    // the lane's 8 reference pixels of the chosen patch
    unsigned r_lo, r_hi;
    {
        int wx = mbx + cx - winx0 + pcol, wy = mby + cy - winy0 + prow;
        const unsigned *d = reinterpret_cast<const unsigned *>(win + wy * kRefStride + (wx & ~3));
        unsigned d0 = d[0], d1 = d[1], d2 = d[2];
        r_lo = __builtin_amdgcn_alignbyte(d1, d0, wx & 3);
        r_hi = __builtin_amdgcn_alignbyte(d2, d1, wx & 3);
    }
    uint2 srcpx = *reinterpret_cast<const uint2 *>(tile + prow * 128 + mbl * 16 + pcol);
    int outpx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) outpx[k] = byte_of(k < 4 ? r_lo : r_hi, k & 3);   // skip: copy the patch (:281-283)

    const bool wave_coded = __any(coded);
    if (wave_coded) {   // wavefront-uniform branch: the LDS transposes need all lanes
        LaneQ lq;
        load_lane_q<true>(lq, qtabs + p.qsel, i);
        int v[8], qc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int d = byte_of(k < 4 ? srcpx.x : srcpx.y, k & 3) - outpx[k];   // calc_residuals (:118-119); |d| <= 255
            v[k] = (int)((unsigned)tdiv2(d) << 8);                          // (:304)
        }
        int *sub = scratch + wave * kScratchDwords + (h * 4 + s) * kSubStride;
        forward_subblock(v, sub, i, lq, qc, stage_sub);
        if (recon) {
            int dpx[8];
            inverse_subblock(qc, sub, i, lq, dpx);
            if (coded) {
#pragma unroll
                for (int k = 0; k < 8; k++)   // apply_residuals (:98-104)
                    outpx[k] = min(max(outpx[k] + (dpx[k] - 128) * 2, 0), 255);
            }
        }
    }
    wave_lds_sync();   // a skipped macroblock's lanes overwrite whatever the wavefront-wide transform scattered
    if (!coded) {   // zero coefficients for skipped macroblocks (API contract; the reference has None)
#pragma unroll
        for (int k = 0; k < 8; k++) stage_sub[i * 8 + k] = 0;
    }
    __syncthreads();   // all search / residual reads of `tile` are done
    if (recon) *reinterpret_cast<uint2 *>(tile + prow * 128 + mbl * 16 + pcol) = pack8(outpx);
    __syncthreads();

    store_coef_strip(stage, coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256, sp.n_mb);
    if (recon) store_recon_strip(tile, recon + (long)sp.stream * g.pad_frame_bytes + p.pad_off, p, sp.x0, sp.y0, sp.n_mb);
}

// ================================================================== I-frame decode
// reference: VideoPlane::decode_plane / decode_plane_into (src/common.rs:423-446, 477-496)
__global__ __launch_bounds__(kThreads) void k_dec_iframe(FrameGeom g, const int16_t *__restrict__ coef,
                                                          uint8_t *__restrict__ out, const QTab *__restrict__ qtabs)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kCoefBytes];
    __shared__ __attribute__((aligned(16))) int scratch[4 * kScratchDwords];

    const StripPos sp = locate_strip(g);
    const PlaneGeom &p = g.p[sp.plane];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31, s = l >> 3, i = l & 7;
    const int mbl = wave * 2 + h;

    LaneQ lq;
    load_lane_q<false>(lq, qtabs + p.qsel, i);
    load_coef_strip(stage, coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256, sp.n_mb);
    __syncthreads();

    const int16_t *stage_sub = reinterpret_cast<const int16_t *>(stage) + mbl * 256 + s * 64;
    int qc[8], px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qc[k] = (int)stage_sub[lq.zz[k]];
    int *sub = scratch + wave * kScratchDwords + (h * 4 + s) * kSubStride;
    inverse_subblock(qc, sub, i, lq, px);
    const int prow = (s >> 1) * 8 + i, pcol = mbl * 16 + (s & 1) * 8;
    *reinterpret_cast<uint2 *>(tile + prow * 128 + pcol) = pack8(px);
    __syncthreads();
    store_recon_strip(tile, out + (long)sp.stream * g.pad_frame_bytes + p.pad_off, p, sp.x0, sp.y0, sp.n_mb);
}

// ================================================================== P-frame decode
// reference: VideoPlane::decode_plane_delta / _into (src/common.rs:448-475, 498-521) ->
// decode_block_delta (:254-285).  ref and out are distinct buffers (ping-pong): the
// reference reads every patch from the old plane before it writes anything (:498-521).
// err_flag is set when a motion vector leaves the plane (:258-259 debug_assert); the
// vector is then treated as (0,0) so that no out-of-bounds access happens.
__global__ __launch_bounds__(kThreads) void k_dec_pframe(FrameGeom g, const int8_t *__restrict__ mv,
                                                          const uint8_t *__restrict__ has, const int16_t *__restrict__ coef,
                                                          const uint8_t *__restrict__ ref, uint8_t *__restrict__ out,
                                                          const QTab *__restrict__ qtabs, int *__restrict__ err_flag)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kCoefBytes];
    __shared__ __attribute__((aligned(16))) int scratch[4 * kScratchDwords];

    const StripPos sp = locate_strip(g);
    const PlaneGeom &p = g.p[sp.plane];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31, s = l >> 3, i = l & 7;
    const int mbl = wave * 2 + h;
    const bool mb_valid = mbl < sp.n_mb;
    const long mbi = (long)sp.stream * g.mbs_per_frame + sp.mb_first + (mb_valid ? mbl : 0);

    load_coef_strip(stage, coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256, sp.n_mb);

    int mx = mv[mbi * 2 + 0], my = mv[mbi * 2 + 1];
    const bool coded = mb_valid && has[mbi] != 0;
    const int mbx = sp.x0 + mbl * 16, mby = sp.y0;
    if (mb_valid) {
        int sx = mbx + mx, sy = mby + my;
        if (sx < 0 || sx > p.pw - 16 || sy < 0 || sy > p.ph - 16) {
            if (l == 0) atomicOr(err_flag, 1);
            mx = 0; my = 0;
        }
    } else {
        mx = 0; my = 0;
    }
    const int prow = (s >> 1) * 8 + i, pcol = (s & 1) * 8;
    int outpx[8];
    if (mb_valid) {   // the lane's 8 pixels of the motion-compensated patch (get_block, :327-339)
        const uint8_t *refp = ref + (long)sp.stream * g.pad_frame_bytes + p.pad_off;
        long off = (long)(mby + my + prow) * p.pw + (mbx + mx + pcol);
        int sh = (int)(off & 3);
        const unsigned *d = reinterpret_cast<const unsigned *>(refp + (off - sh));
        unsigned d0 = d[0], d1 = d[1], d2 = sh ? d[2] : 0u;
        unsigned lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#pragma unroll
        for (int k = 0; k < 8; k++) outpx[k] = byte_of(k < 4 ? lo : hi, k & 3);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) outpx[k] = 0;
    }
    __syncthreads();   // coefficient stage ready

    if (__any(coded)) {
        LaneQ lq;
        load_lane_q<false>(lq, qtabs + p.qsel, i);
        const int16_t *stage_sub = reinterpret_cast<const int16_t *>(stage) + mbl * 256 + s * 64;
        int qc[8], dpx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) qc[k] = (int)stage_sub[lq.zz[k]];
        int *sub = scratch + wave * kScratchDwords + (h * 4 + s) * kSubStride;
        inverse_subblock(qc, sub, i, lq, dpx);
        if (coded) {
#pragma unroll
            for (int k = 0; k < 8; k++) outpx[k] = min(max(outpx[k] + (dpx[k] - 128) * 2, 0), 255);
        }
    }
    *reinterpret_cast<uint2 *>(tile + prow * 128 + mbl * 16 + pcol) = pack8(outpx);
    __syncthreads();
    store_recon_strip(tile, out + (long)sp.stream * g.pad_frame_bytes + p.pad_off, p, sp.x0, sp.y0, sp.n_mb);
}

// ================================================================== plane blits
// reference: VideoPlane::blit (src/plane.rs:20-29), byte-granular rectangle copy.
__global__ __launch_bounds__(kThreads) void k_blit(uint8_t *__restrict__ dst, int dst_w, const uint8_t *__restrict__ src,
                                                    int src_w, int dx, int dy, int sx, int sy, int sw, int sh)
{
    long n = (long)sw * sh;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        int row = (int)(idx / sw), col = (int)(idx - (long)row * sw);
        dst[(long)(row + dy) * dst_w + dx + col] = src[(long)(row + sy) * src_w + sx + col];
    }
}

// Decoder::advance_frame's three crop blits of the padded framebuffer into the unpadded
// retframe (src/dec.rs:195-197, 209-211), all planes and streams in one launch.
// One thread moves 16 bytes when the geometry allows it, else byte by byte.
__global__ __launch_bounds__(kThreads) void k_crop_frames(FrameGeom g, const uint8_t *__restrict__ padded,
                                                           uint8_t *__restrict__ frames)
{
    const int plane = blockIdx.y, stream = blockIdx.z;
    const PlaneGeom &p = g.p[plane];
    const uint8_t *src = padded + (long)stream * g.pad_frame_bytes + p.pad_off;
    uint8_t *dst = frames + (long)stream * g.src_frame_bytes + p.src_off;
    if (p.fast_src) {   // w % 16 == 0 and 16-byte aligned plane bases
        int cpr = p.w >> 4;   // 16-byte chunks per row
        long n = (long)cpr * p.h;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
            int row = (int)(idx / cpr), ch = (int)(idx - (long)row * cpr);
            *reinterpret_cast<uint4 *>(dst + (long)row * p.w + ch * 16) =
                *reinterpret_cast<const uint4 *>(src + (long)row * p.pw + ch * 16);
        }
    } else {
        long n = (long)p.w * p.h;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
            int row = (int)(idx / p.w), col = (int)(idx - (long)row * p.w);
            dst[idx] = src[(long)row * p.pw + col];
        }
    }
}

// VideoFrame::new_padded initial state (src/frame.rs:38-43): Y = 0, U = V = 128.
__global__ __launch_bounds__(kThreads) void k_init_padded(FrameGeom g, uint8_t *__restrict__ padded)
{
    const int plane = blockIdx.y, stream = blockIdx.z;
    const PlaneGeom &p = g.p[plane];
    uint4 *dst = reinterpret_cast<uint4 *>(padded + (long)stream * g.pad_frame_bytes + p.pad_off);
    unsigned f = plane == 0 ? 0u : 0x80808080u;
    long n = ((long)p.pw * p.ph) >> 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        dst[idx] = make_uint4(f, f, f, f);
}

}  // namespace pfv
