// pfv_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels for the Pretty Fast Video
// per-macroblock transform / motion path.  No MFMA: the transforms are 36-add / 12-shift
// integer butterflies with per-stage truncation (reference src/dct.rs:176-293), and the
// motion search is a data-dependent 4-step descent (src/common.rs:154-204).
//
// Work decomposition (all kernels):
//   wavefront  = one STRIP of 8 horizontally adjacent macroblocks (128 x 16 pixels) of one plane of one
//                stream; 8 lanes per macroblock: lane (m, i) = (lane >> 3, lane & 7); the wavefront index is
//                made scalar (readfirstlane) so that all strip geometry lives in SGPRs;
//   workgroup  = 4 wavefronts = 4 strips.  The transform-only kernels never synchronise across wavefronts;
//                the p-frame encoder stacks its 4 strips vertically (a 128 x 64 tile) so that they share one
//                reference window in LDS (two workgroup barriers per tile).
//   transform  : lane (m, i) owns rows i and i+8 of macroblock m and processes them as two passes of two 8x8
//                subblocks (16 values in registers).  The 1-D butterflies run entirely in registers; the 8x8
//                transposes between the row and column passes go through a wavefront-private, XOR-swizzled,
//                pad-free 4 KiB LDS exchange region (conflict-free b128 writes / b32 reads); the same region
//                then serves as the zigzag stage of the coefficient block.  In "column layout" the lane owns
//                column i of its subblocks, so the per-column quantiser constants are shared; they are read
//                from an LDS copy of the 64-entry tables at the point of use.
//   search     : "row owner" form -- lane i keeps source rows i and i+8 in registers and accumulates, for every
//                candidate of a level, sum b^2 - 2 sum ab over its two rows with v_dot4_u32_u8 from one aligned
//                dword span per reference row (SSD = sum a^2 - 2 sum ab + sum b^2, exact in u32); partial sums
//                are all-reduced over the macroblock's 8 lanes with three DPP steps; levels 8 and 4 need no
//                byte realignment, levels 2 and 1 rebase each span once with v_alignbyte_b32.
//   memory     : plane rows go straight between HBM and registers as 16-byte-per-lane vectors that tile
//                128-byte row segments; the reference window is staged with direct global->LDS loads
//                (global_load_lds_dwordx4); the strip's contiguous 4 KiB coefficient block moves as 256-byte
//                runs, 16 B per lane, non-temporal.
//
// Bit-exactness notes (reference line numbers in the function comments):
//   - i32 arithmetic wraps (Rust release) -> done in unsigned here;
//   - `/` truncates toward zero, `>>` is arithmetic;
//   - encode = rows then columns, decode = columns then rows;
//   - encode indexes SCALE/q by raster index, decode by zigzag position (QTab::deq).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfv_device.h"
#include "pfv_prof.h"   // KMARK: phase timestamps of the experiment builds, empty otherwise


namespace pfv {

// reference src/dct.rs:4-13 (data).  Internal linkage: the p-frame encoder is its own translation unit in the product build
// (pfv_penc.hip) and carries its own copy of the two tables
namespace {
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
}  // namespace

// ------------------------------------------------------------------ LDS layout
constexpr int kWinRows = 16 * kStripsPerWG + 30;   // reference window of a 128 x 64 tile: +-15 rows
constexpr int kWinStride = 176;                    // bytes per window row: 160 used (x0-16 .. x0+143)
constexpr int kWinBytes = kWinRows * kWinStride;
constexpr int kWinChunksPerRow = kWinStride / 16;                                    // 11 (10 of pixels + 1 pad)
constexpr int kWinIssues = (kWinRows * kWinChunksPerRow + 63) / 64;                  // 17 wave-wide 1 KiB LDS-DMA loads per window
constexpr int kWinAlloc = kWinIssues * 1024;                                         // 17 KiB per window buffer
// wavefront w issues (and later owns as its exchange region) the contiguous loads [first(w), first(w+1)): 5,4,4,4
__device__ __forceinline__ constexpr int win_first_issue(int w) { return (w * kWinIssues + kStripsPerWG - 1) / kStripsPerWG; }
static_assert(win_first_issue(kStripsPerWG) == kWinIssues, "window issue split");
constexpr int kMBPitch = 2 * 64;                   // dwords per macroblock in the exchange region (bank spread by XOR swizzle)
// The zigzag stage (the exchange region's second use): 64 dwords of coefficients per 8-lane slot, slots kStagePitch dwords apart.
// 16-bit scatters / gathers are serviced 32 lanes = 4 slots at a time on 32 banks; at a pitch of 64 the four slots' identical
// zigzag patterns fell on identical banks (4-way conflict on every access); 72 = 8 (mod 32) moves them 8 banks apart, and a row
// of zigzag positions spans at most 16 dwords.
constexpr int kStagePitch = 72;
constexpr int kStageChunks = kStagePitch / 4;      // 16-byte chunks per slot: 16 of coefficients + 2 of padding
constexpr int kXchgDwords = kStripMB * kMBPitch;   // per wavefront: 1024 dwords = 4 KiB
static_assert(kXchgDwords * 4 <= 4 * 1024, "exchange region must fit a wavefront's smallest window slice");
static_assert(kStripMB * kStagePitch <= kXchgDwords && kStagePitch % 4 == 0, "padded stage fits the exchange region, chunks stay 16-byte aligned");

// ------------------------------------------------------------------ small helpers
__host__ __device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__host__ __device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__host__ __device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
// The same wrapping product for operands known to fit 24 signed bits (v_mul_i32_i24 instead of v_mul_lo_u32): transform
// outputs of 8-bit pixels (|m| < 2^22) times DCT_SCALE_FACTOR (<= 43), and i16 coefficients times SCALE*q (< 2^22 for
// q <= 65535, which make_qtab enforces).
__device__ __forceinline__ int wmul24(int a, int b) { return __mul24(a, b); }
// High 32 bits of the 48-bit product of two signed 24-bit operands: ONE v_mul_hi_i32_i24 (the compiler matches the widened
// product of sign-extended 24-bit values and drops the extensions, the instruction ignores bits 24..31 anyway).
// Used as (m * (SCALE << 16)) >> 32 == (m * SCALE) >> 16 -- the arithmetic shift of src/dct.rs:92 -- for |m| < 2^23,
// SCALE << 16 <= 43 << 16 < 2^23.
__host__ __device__ __forceinline__ int mulhi24(int a, int b)
{
    const int x = (int)((unsigned)a << 8) >> 8, y = (int)((unsigned)b << 8) >> 8;
    return (int)(((long long)x * (long long)y) >> 32);
}

// Rust `/` by 2, 4, 16 on i32 (truncation toward zero).  trunc(x / 2^k) = (x + bias) >> k with
// bias = (2^k - 1) for negative x; truncating divisions compose (trunc(trunc(x/a)/b) = trunc(x/(ab)))
// and keep the sign (or give 0, where the bias is harmless), so x/4 is taken from x/2 and x/16
// from x/4 with the SAME one- or two-bit bias: one sign extraction + one mask per operand.
struct TDiv24 {   // operand needing x/2 and x/4
    int x, h, q;
    __host__ __device__ __forceinline__ explicit TDiv24(int v) : x(v)
    {
        unsigned b = (unsigned)v >> 31;
        h = (int)((unsigned)v + b) >> 1;
        q = (int)((unsigned)h + b) >> 1;
    }
};
struct TDiv416 {   // operand needing x/4 and x/16
    int x, q, s;
    __host__ __device__ __forceinline__ explicit TDiv416(int v) : x(v)
    {
        unsigned b = (unsigned)(v >> 31) & 3u;
        q = (int)((unsigned)v + b) >> 2;
        s = (int)((unsigned)q + b) >> 2;
    }
};
__host__ __device__ __forceinline__ int tdiv2(int x) { return (int)((unsigned)x + ((unsigned)x >> 31)) >> 1; }

// Intra-wavefront LDS hand-off: DS operations of one wavefront execute in issue order, so
// only the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP cross-lane moves
template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kRowHalfMirror = 0x141;   // lane i <-> 7-i inside each 8 lanes

// all-reduce over the 8 lanes of one macroblock
__device__ __forceinline__ int mb_sum(int v)
{
    v += dpp<kQuadXor1>(v);
    v += dpp<kQuadXor2>(v);
    v += dpp<kRowHalfMirror>(v);
    return v;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (observed, speed only).
// Give every XCD one contiguous range of work items so that neighbouring strips -- which
// share reference rows -- meet in the same 4 MiB L2.
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

__device__ __forceinline__ int plane_of(const FrameGeom &g, int idx, bool tiles)
{
    int s1 = tiles ? g.p[1].tile0 : g.p[1].strip0, s2 = tiles ? g.p[2].tile0 : g.p[2].strip0;
    return (g.n_planes > 2 && idx >= s2) ? 2 : ((g.n_planes > 1 && idx >= s1) ? 1 : 0);
}
__device__ __forceinline__ void finish_strip(const FrameGeom &g, StripPos &s)
{
    const PlaneGeom &p = g.p[s.plane];
    s.x0 = s.sx * (kStripMB * 16);
    s.y0 = s.by * 16;
    s.n_mb = min(kStripMB, p.bw - s.sx * kStripMB);
    s.mb_first = p.mb0 + s.by * p.bw + s.sx * kStripMB;
}
// strip index (over all streams) -> position; one strip per wavefront
__device__ __forceinline__ StripPos locate_strip(const FrameGeom &g, int gstrip)
{
    StripPos s;
    s.stream = gstrip / g.strips_per_frame;
    int st = gstrip - s.stream * g.strips_per_frame;
    s.plane = plane_of(g, st, false);
    const PlaneGeom &p = g.p[s.plane];
    st -= p.strip0;
    s.by = st / p.strips_x;
    s.sx = st - s.by * p.strips_x;
    finish_strip(g, s);
    return s;
}

// ------------------------------------------------------------------ 1-D transforms
// reference src/dct.rs:176-239  DctMatrix8x8::fdct -- EXACT form.
// Every forward transform in this codec starts from samples with 8 zero fraction bits
// ((px - 128) << 8, src/common.rs:291, or (delta / 2) << 8, :304).  In the row pass all operands of the
// truncating divisions (/2, /4, /16, dct.rs:206-214) are then multiples of 256, so the divisions are exact
// and every row output is a multiple of 16; in the column pass the operands are multiples of 16, so the
// divisions are exact again.  Truncation toward zero therefore never happens in an fdct, and `x / 2^k` is a
// plain arithmetic shift -- no sign bias needed (tests/test_oracle.py::test_forward_dct_is_exact checks the
// claim against the truncating form; the GPU parity tests check the kernels bit for bit).
__host__ __device__ __forceinline__ void fdct8(int (&v)[8])
{
    int a0 = wadd(v[0], v[7]), a1 = wadd(v[1], v[6]), a2 = wadd(v[2], v[5]), a3 = wadd(v[3], v[4]);
    int a4 = wsub(v[0], v[7]), a5 = wsub(v[1], v[6]), a6 = wsub(v[2], v[5]), a7 = wsub(v[3], v[4]);
    int b0 = wadd(a0, a3), b1 = wadd(a1, a2), b2 = wsub(a0, a3), b3 = wsub(a1, a2);
    int c0 = wadd(b0, b1), c1 = wsub(b0, b1);
    int c2 = wadd(wadd(b2, b2 >> 2), b3 >> 1);
    int c3 = wsub(wsub(b2 >> 1, b3), b3 >> 2);
    int a4q = a4 >> 2, a7q = a7 >> 2;
    int b4 = wsub(wadd(wadd(a7q, a4), a4q), a4 >> 4);
    int b7 = wadd(wsub(wsub(a4q, a7), a7q), a7 >> 4);
    int b5 = wsub(wsub(wadd(a5, a6), a6 >> 2), a6 >> 4);
    int b6 = wadd(wadd(wsub(a6, a5), a5 >> 2), a5 >> 4);
    int c4 = wadd(b4, b5), c5 = wsub(b4, b5), c6 = wadd(b6, b7), c7 = wsub(b6, b7);
    v[0] = c0; v[1] = c4; v[2] = c2; v[3] = wsub(c5, c7);
    v[4] = c1; v[5] = wadd(c5, c7); v[6] = c3; v[7] = c6;
}

// reference src/dct.rs:241-293  DctMatrix8x8::idct
__host__ __device__ __forceinline__ void idct8(int (&v)[8])
{
    int c0 = v[0], d4 = v[1], d6 = v[3], c1 = v[4], d5 = v[5], d7 = v[7];
    TDiv24 c2(v[2]), c3(v[6]);
    int c4 = d4, c5 = wadd(d5, d6), c7 = wsub(d5, d6), c6 = d7;
    TDiv416 b4(wadd(c4, c5)), b5(wsub(c4, c5)), b6(wadd(c6, c7)), b7(wsub(c6, c7));
    int b0 = wadd(c0, c1), b1 = wsub(c0, c1);
    int b2 = wadd(wadd(c2.x, c2.q), c3.h);
    int b3 = wsub(wsub(c2.h, c3.x), c3.q);
    int a4 = wsub(wadd(wadd(b7.q, b4.x), b4.q), b4.s);
    int a7 = wadd(wsub(wsub(b4.q, b7.x), b7.q), b7.s);
    int a5 = wadd(wadd(wsub(b5.x, b6.x), b6.q), b6.s);
    int a6 = wsub(wsub(wadd(b6.x, b5.x), b5.q), b5.s);
    int a0 = wadd(b0, b2), a1 = wadd(b1, b3), a2 = wsub(b1, b3), a3 = wsub(b0, b2);
    v[0] = wadd(a0, a4); v[1] = wadd(a1, a5); v[2] = wadd(a2, a6); v[3] = wadd(a3, a7);
    v[4] = wsub(a3, a7); v[5] = wsub(a2, a6); v[6] = wsub(a1, a5); v[7] = wsub(a0, a4);
}

// ------------------------------------------------------------------ 8x8 transposes through LDS
// Exchange region of macroblock m: two subblocks of 64 dwords, M[s][row][col], XOR-swizzled so that neither
// direction has bank conflicts without any padding:
//   * the two 16-byte halves of a row are swapped when (row & 4): lanes i and i+4 of a b128 write (serviced in
//     groups of 8 consecutive lanes = one macroblock) then hit different banks;
//   * row `row` of macroblock m is stored at row position row ^ (m & 3): in a column read (b32, serviced in groups
//     of 32 lanes = 4 macroblocks whose regions are 128 dwords = 0 banks apart) the 4 macroblocks then read 4
//     different rows, i.e. banks 8 apart.
// row layout (lane i holds M[s][i][0..7])  ->  column layout (lane i holds M[s][0..7][i])
__device__ __forceinline__ void rows_to_cols2(int (&v)[2][8], int *mb, int i, int mx)
{
    const int sw = (i >> 2) & 1;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        int4 *w = reinterpret_cast<int4 *>(mb + s * 64 + (i ^ mx) * 8);
        w[sw] = make_int4(v[s][0], v[s][1], v[s][2], v[s][3]);
        w[sw ^ 1] = make_int4(v[s][4], v[s][5], v[s][6], v[s][7]);
    }
    wave_lds_sync();
    const int lo = i, hi = i ^ 4;
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int r = 0; r < 8; r++) v[s][r] = mb[s * 64 + (r ^ mx) * 8 + (r < 4 ? lo : hi)];
    }
    wave_lds_sync();
}
// column layout  ->  row layout
__device__ __forceinline__ void cols_to_rows2(int (&v)[2][8], int *mb, int i, int mx)
{
    const int lo = i, hi = i ^ 4;
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int r = 0; r < 8; r++) mb[s * 64 + (r ^ mx) * 8 + (r < 4 ? lo : hi)] = v[s][r];
    }
    wave_lds_sync();
    const int sw = (i >> 2) & 1;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int4 *w = reinterpret_cast<const int4 *>(mb + s * 64 + (i ^ mx) * 8);
        int4 a = w[sw], b = w[sw ^ 1];
        v[s][0] = a.x; v[s][1] = a.y; v[s][2] = a.z; v[s][3] = a.w;
        v[s][4] = b.x; v[s][5] = b.y; v[s][6] = b.z; v[s][7] = b.w;
    }
    wave_lds_sync();
}

// Per-lane quantiser constants for column c of a subblock (raster indices k*8 + c), shared by the lane's
// four subblocks.  They are read from the LDS copy of the tables at the point of use (8 distinct addresses per
// wave-instruction: conflict-free broadcast), which keeps them out of the long-lived register set.
// One 16-byte entry per raster index, {rcp, scale, zigzag position, deq}: the quantiser's three constants arrive with one
// ds_read_b128 and the float reciprocal sits in the LOW half of an aligned register pair, which is where a packed-f32
// instruction broadcasts an operand from (the separate-table layout of round 2 cost a v_mov_b32 per pair to get it there).
struct LaneQ {
    const int *tab;   // LDS table, see fill_qtable
    int c;            // the lane's column
    __device__ __forceinline__ float rcp(int k) const { return __int_as_float(tab[(k * 8 + c) * 4 + 0]); }   // biased 1/q (encode)
    __device__ __forceinline__ int scale(int k) const { return tab[(k * 8 + c) * 4 + 1]; }    // DCT_SCALE_FACTOR[k*8+c] (encode); << 16 in the float encoders (fill_qtable<true, true>)
    __device__ __forceinline__ int zz(int k) const { return tab[(k * 8 + c) * 4 + 2]; }       // INV_ZIGZAG[k*8+c]
    __device__ __forceinline__ int deq(int k) const { return tab[(k * 8 + c) * 4 + 3]; }      // SCALE[z]*q[z], z = zz (decode)
    __device__ __forceinline__ float deqf(int k) const { return __int_as_float(tab[(k * 8 + c) * 4 + 3]); }   // the same as f32 bits (fill_qtable<true, true>)
};
// The 64-entry tables live in LDS (kQTabDwords per copy), written once per wavefront / workgroup with coalesced loads and one
// 16-byte store per lane, so that the per-lane constants cost LDS reads instead of 32 scattered vector memory loads per lane
// (which would put more bytes through the CU's texture-addresser path than the pixels and coefficients themselves).
constexpr int kQTabDwords = 4 * 64;
template <bool ENC, bool FLT = false>
__device__ __forceinline__ int4 load_qentry(const QTab *qt, int lane)   // this lane's table entry (two coalesced global loads)
{
    int4 e;
    e.x = ENC ? __float_as_int(qt->rcp[lane]) : 0;
    e.y = ENC ? (FLT ? (kScale[lane] << 16) : kScale[lane]) : 0;   // float encoders: second operand of mulhi24 (quant_scale)
    e.z = kInvZigzag[lane];
    e.w = FLT ? __float_as_int((float)qt->deq[lane]) : qt->deq[lane];   // float form: deq < 2^24 (checked on the host)
    return e;
}
template <bool ENC, bool FLT = false>
__device__ __forceinline__ void fill_qtable(int *tab, const QTab *qt, int lane)
{
    reinterpret_cast<int4 *>(tab)[lane] = load_qentry<ENC, FLT>(qt, lane);
}
__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d)
{
    return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16) | ((unsigned)d << 24);
}
__device__ __forceinline__ int byte_of(unsigned w, int i) { return (int)((w >> (8 * i)) & 0xffu); }

// one 16-pixel macroblock row = subblock 2h (left 8 pixels) and 2h+1 (right 8 pixels)
__device__ __forceinline__ void unpack_row(const uint4 &row, int (&px)[2][8])
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        px[0][k] = byte_of(row.x, k); px[0][k + 4] = byte_of(row.y, k);
        px[1][k] = byte_of(row.z, k); px[1][k + 4] = byte_of(row.w, k);
    }
}
__device__ __forceinline__ uint4 pack_row(const int (&px)[2][8])
{
    return make_uint4(pack4(px[0][0], px[0][1], px[0][2], px[0][3]), pack4(px[0][4], px[0][5], px[0][6], px[0][7]),
                      pack4(px[1][0], px[1][1], px[1][2], px[1][3]), pack4(px[1][4], px[1][5], px[1][6], px[1][7]));
}

// ------------------------------------------------------------------ strip I/O
// streaming (read-once / write-once) global accesses: non-temporal, so that coefficient and retframe traffic does
// not evict the reference planes that neighbouring strips re-read from L2 (measured +1.3 % on the GOP bench)
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    return make_uint4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                      __builtin_nontemporal_load(&p->w));
}
__device__ __forceinline__ void st_stream(uint4 *p, const uint4 &v)
{
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}

// 16 source pixels (x .. x+15, row y) of an unpadded plane with the reference's pad rule
// (src/common.rs:352-356: img_copy.fill(clear); blit(source)).
__device__ __forceinline__ uint4 load_src16(const uint8_t *plane, const PlaneGeom &p, int x, int y)
{
    unsigned fill = (unsigned)p.clear * 0x01010101u;
    uint4 val = make_uint4(fill, fill, fill, fill);
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
    return val;
}

// 16 bytes at an arbitrary byte address of a padded plane (get_block, src/common.rs:327-339)
__device__ __forceinline__ uint4 load_unaligned16(const uint8_t *p)
{
    // 16 bytes at an arbitrary address from aligned dwords; the 5th dword is only touched when
    // the address is misaligned, in which case it holds wanted bytes
    int sh = (int)((uintptr_t)p & 3);
    const unsigned *d = reinterpret_cast<const unsigned *>(p - sh);
    unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = sh ? d[4] : 0u;
    return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                      __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
}

// Decoder::advance_frame's crop of the padded framebuffer into the unpadded retframe
// (src/dec.rs:195-197, 209-211), fused into the decode kernels: the lane's 16 reconstructed pixels
// of row y, columns x..x+15, also go to the tightly packed output plane if they lie inside it.
// Only used when every plane allows 16-byte vector stores (w % 16 == 0, aligned bases); other
// geometries run the separate k_crop_frames pass.
__device__ __forceinline__ void store_cropped16(uint8_t *plane, const PlaneGeom &p, int x, int y, const uint4 &val)
{
    if (y < p.h && x < p.w) st_stream(reinterpret_cast<uint4 *>(plane + (long)y * p.w + x), val);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// Coefficients of one HALF (h = 0: subblocks 0,1; h = 1: subblocks 2,3) of every macroblock of
// the strip <-> LDS stage (8 macroblocks x 256 B).  Global side: 256-byte runs, 16 B per lane.
// 16-byte chunk ch (16 per slot) of the padded stage
__device__ __forceinline__ int stage_chunk(int ch) { return (ch >> 4) * kStageChunks + (ch & 15); }
__device__ __forceinline__ void store_coef_half(const int *stage, int16_t *coef_mb0, int n_mb, int lane, int h)
{
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int ch = j * 64 + lane;   // 16-byte chunk of the stage; 16 chunks per macroblock half
        int mb = ch >> 4;
        if (mb < n_mb) st_stream(&reinterpret_cast<uint4 *>(coef_mb0)[mb * 32 + h * 16 + (ch & 15)], reinterpret_cast<const uint4 *>(stage)[stage_chunk(ch)]);
    }
}
__device__ __forceinline__ void fetch_coef_half(uint4 (&buf)[2], const int16_t *coef_mb0, int n_mb, int lane, int h)
{
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int ch = j * 64 + lane;
        int mb = ch >> 4;
        buf[j] = make_uint4(0, 0, 0, 0);
        if (mb < n_mb) buf[j] = ld_stream(&reinterpret_cast<const uint4 *>(coef_mb0)[mb * 32 + h * 16 + (ch & 15)]);
    }
}
// The same for the small-grid lane mapping (16 lanes per macroblock, see "Lane mappings" below): the wavefront's 8 slots are the
// two halves of 4 macroblocks -- slot s = macroblock (half_strip * 4 + s / 2), half s % 2 -- so the stage is 4 WHOLE macroblocks,
// 2 KiB contiguous in the coefficient buffer.
__device__ __forceinline__ void store_coef_quads(const int *stage, int16_t *coef_mb0, int n_mb, int lane, int half_strip)
{
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int ch = j * 64 + lane;
        int mb = half_strip * 4 + (ch >> 5);
        if (mb < n_mb) st_stream(&reinterpret_cast<uint4 *>(coef_mb0)[half_strip * 128 + ch], reinterpret_cast<const uint4 *>(stage)[stage_chunk(ch)]);
    }
}
__device__ __forceinline__ void fetch_coef_quads(uint4 (&buf)[2], const int16_t *coef_mb0, int n_mb, int lane, int half_strip)
{
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int ch = j * 64 + lane;
        int mb = half_strip * 4 + (ch >> 5);
        buf[j] = make_uint4(0, 0, 0, 0);
        if (mb < n_mb) buf[j] = ld_stream(&reinterpret_cast<const uint4 *>(coef_mb0)[half_strip * 128 + ch]);
    }
}
__device__ __forceinline__ void stage_coef_half(int *stage, const uint4 (&buf)[2], int lane)
{
#pragma unroll
    for (int j = 0; j < 2; j++) reinterpret_cast<uint4 *>(stage)[stage_chunk(j * 64 + lane)] = buf[j];
}

// ------------------------------------------------------------------ coefficient lists -> zigzag stage (pfv_device.h: CoefLists)
// The decoders' other source of coefficients: the wavefront's macroblocks own one contiguous span [lo, hi) of their stream's list
// (entries ascending by macroblock and position; lo, hi from the frame's exclusive counts).  The stage is cleared and the span's entries are scattered into it with ds_write_b16 --
// what the reference's run loop does per macroblock (src/dec.rs:261-296, 378-417: `coefficients[out_idx] = coeff` onto zeros), 64
// entries per step.  8 lanes per macroblock: the stage holds one HALF of 8 macroblocks per pass (slot = macroblock, entries of the other
// half are passed over); 16 lanes: both halves of 4 macroblocks (slot = 2 x macroblock + half).  `mb_low`: the low 8 bits of the frame-
// relative index of the wavefront's first macroblock -- an entry names its macroblock by those bits, and a wavefront spans at most 8.
// Entries that name a macroblock outside the wavefront's (a list that does not belong to these ranges) are dropped, not written.
struct ListSpan {
    const uint32_t *ent;   // the stream's list
    uint32_t lo, hi;       // the wavefront's span (lo == hi: no entries)
    uint32_t e0, e1;       // entries lo + lane and lo + 64 + lane, fetched ahead (most p-frame strips hold no more)
};
// lo / hi: counts[first macroblock of the wavefront] and counts[one behind its last] -- every lane passes its macroblock's pair (lanes of
// absent macroblocks: anything): lane 0 holds the first macroblock's begin, lane `last_lane` the last one's end
__device__ __forceinline__ ListSpan list_span(const uint32_t *ent, uint32_t begin, uint32_t end, int last_lane, int lane)
{
    ListSpan sp{ent, 0u, 0u, 0u, 0u};
    sp.lo = (uint32_t)__builtin_amdgcn_readlane((int)begin, 0);
    sp.hi = (uint32_t)__builtin_amdgcn_readlane((int)end, last_lane);
    if (sp.hi < sp.lo) sp.hi = sp.lo;
    const uint32_t k = sp.lo + (uint32_t)lane;
    if (k < sp.hi) sp.e0 = ent[k];
    if (k + 64u < sp.hi) sp.e1 = ent[k + 64u];
    return sp;
}
template <int LPM>
__device__ __forceinline__ void stage_list_put(int16_t *stage, uint32_t e, int mb_low, int h)
{
    const int pos = (int)(e & 255u), mbl = (int)(((e >> 8) - (uint32_t)mb_low) & 255u);
    if (LPM == 8) {
        if ((pos >> 7) == h && mbl < kStripMB) stage[mbl * (2 * kStagePitch) + (pos & 127)] = (int16_t)(e >> 16);
    } else {
        if (mbl < kStripMB / 2) stage[(mbl * 2 + (pos >> 7)) * (2 * kStagePitch) + (pos & 127)] = (int16_t)(e >> 16);
    }
}
template <int LPM>
__device__ __forceinline__ void stage_list_half(int *xw, const ListSpan &sp, int lane, int mb_low, int h)
{
#pragma unroll
    for (int j = 0; j < 2; j++) reinterpret_cast<uint4 *>(xw)[stage_chunk(j * 64 + lane)] = make_uint4(0, 0, 0, 0);
    wave_lds_sync();
    int16_t *stage = reinterpret_cast<int16_t *>(xw);
    const uint32_t k = sp.lo + (uint32_t)lane;
    if (k < sp.hi) stage_list_put<LPM>(stage, sp.e0, mb_low, h);
    if (k + 64u < sp.hi) stage_list_put<LPM>(stage, sp.e1, mb_low, h);
    for (uint32_t j = k + 128u; j < sp.hi; j += 64u) stage_list_put<LPM>(stage, sp.ent[j], mb_low, h);
}

// ------------------------------------------------------------------ half-macroblock pipelines (per lane: 2 subblocks)
// Forward: row-layout inputs (24.8 fixed point) -> quantised coefficients in column layout
// (left in v) and scattered in zigzag order into the strip's coefficient stage.
// reference src/common.rs:294-297 + src/dct.rs:88-99 (encode: n = (m*SCALE)>>16; n / q).
// The division is float(n) * rcp followed by a truncating convert; exact for |n| <= 2^15
// (QTab::rcp); |n| <= 5160 for any u8 input (|m| <= 240 * 32768, SCALE <= 43).
// The reference's `as i16` (src/dct.rs:95) cannot wrap for these magnitudes, so it is a no-op here.
// `keep`: lanes of skipped / absent macroblocks store zeros instead.
__device__ __forceinline__ void forward_half(int (&v)[2][8], int *xw, int m, int i, const LaneQ &lq, bool keep)
{
    const int keepmask = keep ? -1 : 0;
    int *mb = xw + m * kMBPitch;
    fdct8(v[0]);   // dct_transform_rows
    fdct8(v[1]);
    rows_to_cols2(v, mb, i, m & 3);
    fdct8(v[0]);   // dct_transform_columns
    fdct8(v[1]);
    int16_t *stage = reinterpret_cast<int16_t *>(xw) + m * (2 * kStagePitch);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int scale = lq.scale(k), zz = lq.zz(k);
        const float rcp = lq.rcp(k);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            int n = wmul24(v[s][k], scale) >> 16;
            v[s][k] = (int)((float)n * rcp) & keepmask;
            stage[s * 64 + zz] = (int16_t)v[s][k];
        }
    }
    wave_lds_sync();
}
// Inverse: quantised coefficients in column layout (v) -> row-layout term t = min(x >> 8, 127).
// The reference's pixel is ((x >> 8) + 128).clamp(0, 255) (src/common.rs:321); callers apply the
// +128 and the clamp.  For p-frames the reference clamps that "delta byte" d first and then forms
// clamp(prev + (d - 128) * 2) (src/common.rs:100-102), i.e. clamp(prev + 2 * clamp(t, -128, 127)).
// The lower clamp of t is redundant (prev <= 255, so any t <= -128 saturates to 0 either way); the
// upper one is not (t = 127 gives prev + 254, t = 128 would give 255 for prev = 0), so it is kept.
// |t| < 2^23, so 2t cannot overflow.
// reference src/dct.rs:75-86 (decode) + src/common.rs:314-316 (columns first, then rows).
__device__ __forceinline__ void inverse_half(int (&v)[2][8], int *xw, int m, int i, const LaneQ &lq)
{
    int *mb = xw + m * kMBPitch;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int deq = lq.deq(k);
        v[0][k] = wmul24(v[0][k], deq);
        v[1][k] = wmul24(v[1][k], deq);
    }
    idct8(v[0]);   // dct_inverse_transform_columns
    idct8(v[1]);
    cols_to_rows2(v, mb, i, m & 3);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        idct8(v[s]);   // dct_inverse_transform_rows
#pragma unroll
        for (int k = 0; k < 8; k++) v[s][k] = min(v[s][k] >> 8, 127);
    }
}
// gather the lane's column of quantised coefficients out of the zigzag-ordered stage
__device__ __forceinline__ void gather_half(int (&v)[2][8], const int *xw, int m, const LaneQ &lq)
{
    const int16_t *stage = reinterpret_cast<const int16_t *>(xw) + m * (2 * kStagePitch);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int zz = lq.zz(k);
        v[0][k] = (int)stage[zz];
        v[1][k] = (int)stage[64 + zz];
    }
    wave_lds_sync();
}

// ------------------------------------------------------------------ the same transforms in f32 (encoders only)
// Inside the ENCODERS every value of the forward transform, of the dequantised coefficients and of the inverse transform
// is an integer far below 2^24: |fdct| <= 2.5 M for 8-bit pixels / residuals whatever the quantiser, and for the session's
// quality-derived tables the inverse stays below 1.9 M even by an L1 worst-case bound over all 64 coefficients at once
// (tests/test_float_exact.py proves both; pfv_enc_session_create re-checks the bound for the tables it is given and
// falls back to the integer kernels otherwise).  Integers below 2^24 and their products with 1/2, 1/4, 5/4, 11/16, 19/16 --
// exact because the operands are multiples of 16 where the integer form divides by 16 -- are exact in f32, so the float
// butterflies below produce THE SAME numbers as fdct8 / idct8, with two gains on gfx950 where every VALU instruction of a
// mixed stream costs ~4.2 cycles: (1) v_pk_add/mul/fma_f32 work on two values (the lane's two subblocks) per
// instruction, (2) fma fuses the shift-and-add pairs of the integer butterfly (30 packed instructions for two 8-point
// forward transforms instead of 2 x 48), and truncation toward zero is one v_trunc_f32 instead of a sign-bias sequence
// (72 instead of 2 x 70 for two inverse transforms).  The decoders keep the integer form: they must wrap exactly like i32
// on hostile coefficients.
typedef float f2 __attribute__((vector_size(8)));   // lane's two subblocks, element s = subblock s
__device__ __forceinline__ f2 f2s(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 f2trunc(f2 x) { return f2{__builtin_truncf(x[0]), __builtin_truncf(x[1])}; }
__device__ __forceinline__ f2 f2floor(f2 x) { return f2{__builtin_floorf(x[0]), __builtin_floorf(x[1])}; }

// src/dct.rs:176-239 in exact f32 (see fdct8)
__device__ __forceinline__ void ffdct8(f2 (&v)[8])
{
    const f2 a0 = v[0] + v[7], a1 = v[1] + v[6], a2 = v[2] + v[5], a3 = v[3] + v[4];
    const f2 a4 = v[0] - v[7], a5 = v[1] - v[6], a6 = v[2] - v[5], a7 = v[3] - v[4];
    const f2 b0 = a0 + a3, b1 = a1 + a2, b2 = a0 - a3, b3 = a1 - a2;
    const f2 c0 = b0 + b1, c1 = b0 - b1;
    const f2 c2 = b2 * f2s(1.25f) + b3 * f2s(0.5f);             // b2 + b2/4 + b3/2
    const f2 c3 = b2 * f2s(0.5f) - b3 * f2s(1.25f);             // b2/2 - b3 - b3/4
    const f2 b4 = a4 * f2s(1.1875f) + a7 * f2s(0.25f);          // a7/4 + a4 + a4/4 - a4/16
    const f2 b7 = a4 * f2s(0.25f) - a7 * f2s(1.1875f);          // a4/4 - a7 - a7/4 + a7/16
    const f2 b5 = a5 + a6 * f2s(0.6875f);                       // a5 + a6 - a6/4 - a6/16
    const f2 b6 = a6 - a5 * f2s(0.6875f);                       // a6 - a5 + a5/4 + a5/16
    const f2 c4 = b4 + b5, c5 = b4 - b5, c6 = b6 + b7, c7 = b6 - b7;
    v[0] = c0; v[1] = c4; v[2] = c2; v[3] = c5 - c7;
    v[4] = c1; v[5] = c5 + c7; v[6] = c3; v[7] = c6;
}
// src/dct.rs:241-293 in exact f32 (see idct8): x / 2^k of the reference = truncation toward zero = v_trunc_f32 of the exact quotient
__device__ __forceinline__ void fidct8(f2 (&v)[8])
{
    const f2 c0 = v[0], d4 = v[1], c2 = v[2], d6 = v[3], c1 = v[4], d5 = v[5], c3 = v[6], d7 = v[7];
    const f2 c5 = d5 + d6, c7 = d5 - d6;
    const f2 b4 = d4 + c5, b5 = d4 - c5, b6 = d7 + c7, b7 = d7 - c7;
    const f2 b0 = c0 + c1, b1 = c0 - c1;
    const f2 half = f2s(0.5f), quarter = f2s(0.25f);
    const f2 c2h = f2trunc(c2 * half), c2q = f2trunc(c2h * half), c3h = f2trunc(c3 * half), c3q = f2trunc(c3h * half);
    const f2 b4q = f2trunc(b4 * quarter), b4s = f2trunc(b4q * quarter), b5q = f2trunc(b5 * quarter), b5s = f2trunc(b5q * quarter);
    const f2 b6q = f2trunc(b6 * quarter), b6s = f2trunc(b6q * quarter), b7q = f2trunc(b7 * quarter), b7s = f2trunc(b7q * quarter);
    const f2 b2 = c2 + c2q + c3h, b3 = c2h - c3 - c3q;
    const f2 a4 = b7q + b4 + b4q - b4s;
    const f2 a7 = b4q - b7 - b7q + b7s;
    const f2 a5 = b5 - b6 + b6q + b6s;
    const f2 a6 = b6 + b5 - b5q - b5s;
    const f2 a0 = b0 + b2, a1 = b1 + b3, a2 = b1 - b3, a3 = b0 - b2;
    v[0] = a0 + a4; v[1] = a1 + a5; v[2] = a2 + a6; v[3] = a3 + a7;
    v[4] = a3 - a7; v[5] = a2 - a6; v[6] = a1 - a5; v[7] = a0 - a4;
}
// The LDS transposes of the float form.  A lane's values are (subblock 0, subblock 1) pairs in aligned register pairs, so the
// exchange region of macroblock m is laid out M[row][col][s] (16 dwords per row, 128 per macroblock, as before) and moves
// pairs: row layout <-> 4 x 16-byte chunks {col 2j, col 2j+1} x {s0, s1}, column layout <-> 8 x 8-byte pairs -- half the LDS
// read instructions of the integer form and no register shuffling.  Conflict-free without padding by two XOR swizzles:
//   * row `row` of macroblock m sits at row position p = row ^ (m & 3): the 4 macroblocks of a 32-lane group (regions 128
//     dwords = 0 banks apart) touch 4 different rows in a column access;
//   * chunk j of the row at position p sits at chunk (j ^ ((p >> 1) & 3)): the 8 lanes of a 16-byte row access (rows 64 bytes
//     apart) land in 8 different 16-byte bank groups.
__device__ __forceinline__ int f_chunk(int p, int j) { return j ^ ((p >> 1) & 3); }
// row layout (lane i holds row i: x[k] = {M0[i][k], M1[i][k]})  ->  column layout (lane i holds column i: x[r] = {M0[r][i], M1[r][i]})
__device__ __forceinline__ void f_rows_to_cols(f2 (&x)[8], int *mb, int i, int mx)
{
    {
        const int p = i ^ mx;
        float4 *row = reinterpret_cast<float4 *>(mb + p * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) row[f_chunk(p, j)] = make_float4(x[2 * j][0], x[2 * j][1], x[2 * j + 1][0], x[2 * j + 1][1]);
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int p = r ^ mx;
        x[r] = *reinterpret_cast<const f2 *>(mb + p * 16 + f_chunk(p, i >> 1) * 4 + (i & 1) * 2);
    }
    wave_lds_sync();
}
// column layout  ->  row layout
__device__ __forceinline__ void f_cols_to_rows(f2 (&x)[8], int *mb, int i, int mx)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int p = r ^ mx;
        *reinterpret_cast<f2 *>(mb + p * 16 + f_chunk(p, i >> 1) * 4 + (i & 1) * 2) = x[r];
    }
    wave_lds_sync();
    {
        const int p = i ^ mx;
        const float4 *row = reinterpret_cast<const float4 *>(mb + p * 16);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 c = row[f_chunk(p, j)];
            x[2 * j] = f2{c.x, c.y};
            x[2 * j + 1] = f2{c.z, c.w};
        }
    }
    wave_lds_sync();
}
// The float encoders' quantiser for one coefficient pair (the lane's two subblocks, same raster position), src/dct.rs:88-99,
// in two steps that csrc/pfv_selfcheck.hip runs on the device for EVERY operand (tests/test_device_selfcheck.py):
//   quant_scale   n = (m * SCALE) >> 16 (dct.rs:92): m = transform output, an exact integer |m| < 2^23 held in f32; scale16 =
//                 DCT_SCALE_FACTOR[idx] << 16.  One v_mul_hi_i32_i24 between two conversions (the product needs 28 bits: not
//                 an f32 operation).
//   quant_div     n / q (truncating, dct.rs:95) = trunc(float(n) * rcp) with rcp = QTab::rcp[idx], the biased reciprocal: one
//                 v_pk_mul_f32 for the pair + two v_trunc_f32.  Returns the quotients as floats (what the closed loop's dequantiser
//                 multiplies next); `biased` = quotient + 1.5 * 2^23 carries the quotient's two's complement in its low mantissa
//                 bits (|quotient| < 2^15), so a 16-bit store of the register's low half IS the reference's `as i16` -- one
//                 packed add for the pair instead of two float -> int conversions.
//   residual_f    trunc(delta / 2) << 8 of a pixel pair (src/common.rs:118-119, :304), see there.
__device__ __forceinline__ f2 quant_scale(f2 m, int scale16)
{
    f2 n;
#pragma unroll
    for (int s = 0; s < 2; s++) n[s] = (float)mulhi24((int)m[s], scale16);
    return n;
}
// magic = 1.5 * 2^23 (kQuantMagic), handed in as a KERNEL ARGUMENT: with the literal the compiler splits the packed add into two
// v_add_f32 (extract-of-binop-with-constant), with an opaque operand it stays one v_pk_add_f32.
constexpr float kQuantMagic = 12582912.0f;
__device__ __forceinline__ f2 quant_div(f2 n, float rcp, float magic, f2 &biased)
{
    const f2 q = f2trunc(n * f2s(rcp));
    biased = q + f2s(magic);
    return q;
}
__device__ __forceinline__ f2 quant_pair_f(f2 m, int scale16, float rcp, float magic, f2 &biased)
{
    return quant_div(quant_scale(m, scale16), rcp, magic, biased);
}
// calc_residuals + the head of encode_subblock_delta (src/common.rs:118-119, :304) for a pixel pair: (delta / 2) << 8 with Rust's
// truncating division, delta = src - prediction in [-255, 255].  t = delta / 2 is a multiple of 1/2 with |t| <= 127.5;
// t * (1 - 2^-10) moves every value less than 1/8 toward zero, so halves leave their tie (toward zero) and integers stay nearest to
// themselves: round-to-nearest of it IS trunc(t).  The rounding is the fma's own: delta * (1/2 - 2^-11) + 1.5 * 2^23 lands on the
// integer grid of [2^23, 2^24) (the product is exact inside the fma, 8 x 11 bits); the second fma takes the magic off and scales by
// 256, exactly (r * 256 and 1.5 * 2^31 are representable, their difference is a small integer).  One v_pk_add_f32 + two v_pk_fma_f32
// per pair, against v_pk_add + v_pk_mul + 2 v_trunc + v_pk_mul.  pfv_selfcheck part 5 runs it for all 511 x 511 (delta, delta) pairs.
__device__ __forceinline__ f2 residual_f(f2 src, f2 pred)
{
    const f2 d = src - pred;
    const float c = 0.5f - 1.0f / 2048.0f;
    const f2 r = f2{__builtin_fmaf(d[0], c, kQuantMagic), __builtin_fmaf(d[1], c, kQuantMagic)};
    return f2{__builtin_fmaf(r[0], 256.0f, -256.0f * kQuantMagic), __builtin_fmaf(r[1], 256.0f, -256.0f * kQuantMagic)};
}
// the 16 bits a ds_write_b16 / a global 16-bit store takes from the biased quotient
__device__ __forceinline__ int16_t quant_low16(float biased) { return (int16_t)__float_as_int(biased); }
// Forward, float form: x = row-layout samples in 24.8 fixed point AS FLOATS ((px - 128) * 256 or trunc(d / 2) * 256); leaves the
// quantised coefficients (column layout, as floats) in x and scatters them in zigzag order into the coefficient stage.
// The quantiser itself is the integer one of forward_half (exact division by reciprocal).
__device__ __forceinline__ void forward_half_f(f2 (&x)[8], int *xw, int m, int i, const LaneQ &lq, float magic)
{
    int *mb = xw + m * kMBPitch;
    ffdct8(x);   // dct_transform_rows (both subblocks)
    f_rows_to_cols(x, mb, i, m & 3);
    ffdct8(x);   // dct_transform_columns
    int16_t *stage = reinterpret_cast<int16_t *>(xw) + m * (2 * kStagePitch);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int zz = lq.zz(k);
        f2 biased;
        x[k] = quant_pair_f(x[k], lq.scale(k), lq.rcp(k), magic, biased);
#pragma unroll
        for (int s = 0; s < 2; s++) stage[s * 64 + zz] = quant_low16(biased[s]);
    }
    wave_lds_sync();
}
// Inverse, float form: c = quantised coefficients in column layout (floats) -> row layout.  lq.deqf = SCALE[z] * q[z] as float
// (fill_qtable<true, true>).
//   PIXEL = false (p-frames): t = floor(x / 256) = x >> 8 as floats; the caller adds the prediction and clamps.
//   PIXEL = true  (i-frames): x / 256 + (127.5 + 1/512), NOT floored: the caller's v_cvt_pk_u8_f32 rounds to nearest even and
//           saturates, and x / 256 is a multiple of 1/256, so the fraction of the sum lies in [0.502, 1.498] above floor(x / 256) + 127
//           and the conversion yields clamp((x >> 8) + 128, 0, 255) -- src/common.rs:321 -- without a v_floor_f32 per pixel
//           (iframe_pixel_f; pfv_selfcheck part 6 runs it for every |x| < 2^24, tools/ubench/cvt_pk_u8_rounding.hip shows the
//           instruction's rounding rule on its own).
constexpr float kPixelBias = 127.501953125f;   // 128 - 1/2 + 1/512
__device__ __forceinline__ f2 iframe_pixel_f(f2 x) { return x * f2s(1.0f / 256.0f) + f2s(kPixelBias); }   // one v_pk_fma_f32; exact wherever the result matters (|x / 256| < 2^15)
template <bool PIXEL>
__device__ __forceinline__ void inverse_half_f(f2 (&c)[8], int *xw, int m, int i, const LaneQ &lq)
{
    int *mb = xw + m * kMBPitch;
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = c[k] * f2s(lq.deqf(k));   // |coefficient * deq| < 2^24: exact
    fidct8(c);   // dct_inverse_transform_columns
    f_cols_to_rows(c, mb, i, m & 3);
    fidct8(c);   // dct_inverse_transform_rows
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = PIXEL ? iframe_pixel_f(c[k]) : f2floor(c[k] * f2s(1.0f / 256.0f));     // v >> 8: |v| < 2^24, exact
}
// 16 reconstructed pixels (floats, integer-valued) -> saturated bytes: v_cvt_pk_u8_f32 clamps to 0..255 and places the byte
__device__ __forceinline__ uint4 pack_row_f(const f2 (&px)[8])
{
    unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        w[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(px[k][0], (unsigned)(k & 3), w[k >> 2]);            // subblock 2h: pixels 0..7
        w[2 + (k >> 2)] = __builtin_amdgcn_cvt_pk_u8_f32(px[k][1], (unsigned)(k & 3), w[2 + (k >> 2)]);   // subblock 2h+1: pixels 8..15
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// the 16 bytes of a row as floats, px[k] = {pixel k, pixel 8 + k}
__device__ __forceinline__ void unpack_row_f(const uint4 &row, f2 (&px)[8])
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        px[k] = f2{(float)byte_of(row.x, k), (float)byte_of(row.z, k)};
        px[k + 4] = f2{(float)byte_of(row.y, k), (float)byte_of(row.w, k)};
    }
}

// ================================================================== I-frame encode (+ closed-loop reconstruction)
// reference: VideoPlane::encode_plane (src/common.rs:351-386) fused with the
// VideoPlane::decode_plane that Encoder::encode_iframe runs on its output
// (src/enc.rs:84-97).  recon == nullptr -> encode only (plane-level operator).
// Lane mappings.  LPM = lanes per macroblock.
//   LPM = 8  (batch mapping, what everything above describes): the wavefront owns a strip of 8 macroblocks; lane (m, i) owns rows
//            i and i + 8 of macroblock m and runs the half-macroblock pipeline twice (h = 0, 1).
//   LPM = 16 (small-grid mapping): the wavefront owns HALF a strip (4 macroblocks); lane = (macroblock, h, i) owns the ONE row
//            8 h + i and runs the pipeline once -- its 8-lane group ("slot" = lane >> 3 = 2 * macroblock + h) is what the
//            transposes, the quantiser tables and the zigzag stage see as "a macroblock's 8 lanes".  Twice the wavefronts, each half
//            as long: when a launch has fewer than a couple of wavefronts per SIMD (one 1080p stream: 1.5), its duration is the
//            serial time of ONE wavefront, and this mapping halves that (the reference's caller is one Encoder per stream,
//            src/enc.rs:125-173).  Same results, bit for bit; the host picks by grid size (pfv_capi.hip: use_small_grid).
template <bool FLT, int LPM = 8>
__global__ __launch_bounds__(kThreads) void k_enc_iframe(FrameGeom g, const uint8_t *__restrict__ src,
                                                          int16_t *__restrict__ coef, uint8_t *__restrict__ recon,
                                                          const QTab *__restrict__ qtabs, float qmagic)
{
    __shared__ __attribute__((aligned(16))) int xchg[kStripsPerWG][kXchgDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kStripsPerWG][kQTabDwords];
    constexpr int kPasses = LPM == 8 ? 2 : 1;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPRs
    const int gw = xcd_remap((int)blockIdx.x, (int)gridDim.x) * kStripsPerWG + wave;
    const int gstrip = LPM == 8 ? gw : gw >> 1, half_strip = LPM == 8 ? 0 : gw & 1;
    if (gstrip >= g.strips_per_frame * g.n_streams) return;   // no cross-wavefront sync in this kernel
    const StripPos sp = locate_strip(g, gstrip);
    if (half_strip * 4 >= sp.n_mb) return;
    const PlaneGeom &p = g.p[sp.plane];
    const int slot = lane >> 3, i = lane & 7;
    const int m = LPM == 8 ? slot : half_strip * 4 + (slot >> 1);   // macroblock within the strip
    int *xw = xchg[wave];

    fill_qtable<true, FLT>(qtab_lds[wave], qtabs + p.qsel, lane);
    const uint8_t *plane = frame_src(g, src, sp.stream) + p.src_off;
    uint4 rows[kPasses];
#pragma unroll
    for (int pass = 0; pass < kPasses; pass++) rows[pass] = load_src16(plane, p, sp.x0 + m * 16, sp.y0 + i + 8 * (LPM == 8 ? pass : (slot & 1)));

    wave_lds_sync();
    const LaneQ lq{qtab_lds[wave], i};
    int16_t *coef_mb0 = coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256;
    uint8_t *dst = recon ? recon + (long)sp.stream * g.pad_frame_bytes + p.pad_off + (long)(sp.y0 + i) * p.pw + sp.x0 + m * 16
                         : nullptr;
#pragma unroll
    for (int pass = 0; pass < kPasses; pass++) {
        const int h = LPM == 8 ? pass : (slot & 1);
        if (FLT) {
            f2 x[8];
            unpack_row_f(rows[pass], x);
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = x[k] * f2s(256.0f) - f2s(32768.0f);   // (px - 128) << 8, src/common.rs:291
            forward_half_f(x, xw, slot, i, lq, qmagic);
            if (LPM == 8) store_coef_half(xw, coef_mb0, sp.n_mb, lane, h);
            else store_coef_quads(xw, coef_mb0, sp.n_mb, lane, half_strip);
            wave_lds_sync();   // the stage has been read back before the region is reused
            if (recon) {
                inverse_half_f<true>(x, xw, slot, i, lq);                             // (v >> 8) + 128, clamped: by the pack's rounding + saturation (src/common.rs:321)
                if (m < sp.n_mb) *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = pack_row_f(x);
            }
        } else {
            int v[2][8];
            unpack_row(rows[pass], v);
#pragma unroll
            for (int s = 0; s < 2; s++) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[s][k] = (int)((unsigned)(v[s][k] - 128) << 8);   // src/common.rs:291
            }
            forward_half(v, xw, slot, i, lq, true);
            if (LPM == 8) store_coef_half(xw, coef_mb0, sp.n_mb, lane, h);
            else store_coef_quads(xw, coef_mb0, sp.n_mb, lane, half_strip);
            wave_lds_sync();   // the stage has been read back before the region is reused
            if (recon) {
                inverse_half(v, xw, slot, i, lq);
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int k = 0; k < 8; k++) v[s][k] = min(max(v[s][k] + 128, 0), 255);   // src/common.rs:321
                if (m < sp.n_mb) *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = pack_row(v);
            }
        }
    }
}

// ================================================================== P-frame encode (+ closed-loop reconstruction)
// reference: VideoPlane::encode_plane_delta (src/common.rs:388-421) -> encode_block_delta
// (:206-236) -> block_search (:154-204) / calc_error (:125-139) / calc_residuals (:108-123)
// / encode_subblock_delta (:300-311), fused with the decode_plane_delta (:448-475,
// decode_block_delta :254-285, apply_residuals :98-104) that Encoder::encode_pframe runs on
// the result (src/enc.rs:134-147).  recon == nullptr -> encode only.
// One workgroup = a 128 x 64 tile (4 vertically stacked strips) sharing one reference window.
//
// Search ("row owner" form): lane i of a macroblock owns source rows i and i+8 for the whole
// search.  For one level it reads, for each vertical candidate offset, the two reference rows
// it needs ONCE as an aligned dword span wide enough for all three horizontal offsets, and
// accumulates  sum b^2 - 2 sum ab  for every candidate from registers; the partial sums are
// all-reduced over the macroblock's 8 lanes with DPP.  After levels 8 and 4 the displacement
// is a multiple of 4 pixels, so those levels need no byte realignment at all; levels 2 and 1
// rebase each span once by (cx & 3) and then use compile-time byte offsets.
struct SearchState {
    int cx, cy;     // accumulated displacement (reference src/common.rs:199-200)
    int err;        // error of the current centre = best so far (:164, :191)
};

// sum a.b over one 16-pixel row (4 dwords)
__device__ __forceinline__ unsigned dot_ab(const uint4 &a, unsigned b0, unsigned b1, unsigned b2, unsigned b3, unsigned acc)
{
    acc = __builtin_amdgcn_udot4(a.x, b0, acc, false);
    acc = __builtin_amdgcn_udot4(a.y, b1, acc, false);
    acc = __builtin_amdgcn_udot4(a.z, b2, acc, false);
    return __builtin_amdgcn_udot4(a.w, b3, acc, false);
}
// sum of squares of two dwords (8 pixels) on top of acc
__device__ __forceinline__ unsigned sq2(unsigned x, unsigned y, unsigned acc)
{
    return __builtin_amdgcn_udot4(y, y, __builtin_amdgcn_udot4(x, x, acc, false), false);
}
__device__ __forceinline__ unsigned sq4(unsigned b0, unsigned b1, unsigned b2, unsigned b3, unsigned acc) { return sq2(b2, b3, sq2(b0, b1, acc)); }

// What one lane of a macroblock needs to finalise "its" candidate of a search level: after the dot products every
// lane holds a partial sum (its two rows) for each of the 8 neighbours; the partials are transposed through a small
// wavefront-private LDS region (8 ds_write_b32 + 2 ds_read_b128 per lane: LDS-pipe work) so that lane c ends up with
// the 8 row-pair partials of candidate c and adds them with plain VALU adds -- instead of all-reducing all 8 values over
// the 8 lanes with 24 DPP adds and building 8 keys in every lane.
constexpr int kPencSplit = -1;        // launch_enc_pframe_kernels: compact_max value that selects the split form (k_pf_search + k_pf_transform)
constexpr int kPsWaves = 7;           // wavefronts per SIMD k_pf_search is compiled for: 69 VGPRs, 17 KiB of LDS per workgroup (register reduce-scatter: no reduction regions)
constexpr int kTfWaves = 7;           // and k_pf_transform: 72 VGPRs, 5.4 KiB of LDS per wavefront
constexpr int kPencCompactMax = 16;   // k_enc_pframe: a tile's coded macroblocks are moved together when there are at most this many (see there)
constexpr int kRedPitch = 72;   // dwords per macroblock: 64 used; 72 = 8 (mod 32) spreads the 4 macroblocks of a 32-lane group over the banks
constexpr int kRedDwords = kStripMB * kRedPitch;
struct SearchLane {
    int neg2;           // -2, handed in as a kernel argument: with an opaque multiplier  bb - 2 ab  stays ONE v_mad_i32_i24 (a literal -2 is
                        // expanded to a shift and a subtract, 33 more instructions per search)
    int ord;            // visiting order 1..8 of this lane's candidate: my outer, mx inner, centre skipped (src/common.rs:168-179)
    int dy, dx;         // its position in the 3 x 3 pattern (-1, 0, 1)
    int *wr;            // &red[m][0][i]: the lane's partial for candidate c goes to wr[c * 8]
    const int4 *rd;     // &red[m][i][0]: the 8 lanes' partials of this lane's candidate
};
__device__ __forceinline__ SearchLane make_search_lane(int *red, int m, int i, int neg2)
{
    SearchLane sl;
    sl.neg2 = neg2;
    sl.ord = i + 1;
    const int b9 = i < 4 ? i : i + 1;           // index in the full 3 x 3 pattern
    sl.dy = b9 / 3 - 1;
    sl.dx = b9 - (b9 / 3) * 3 - 1;
    sl.wr = red + m * kRedPitch + i;
    sl.rd = reinterpret_cast<const int4 *>(red + m * kRedPitch + i * 8);
    return sl;
}
// minimum over the 8 lanes of one macroblock
__device__ __forceinline__ unsigned mb_min(unsigned v)
{
    v = min(v, (unsigned)dpp<kQuadXor1>((int)v));
    v = min(v, (unsigned)dpp<kQuadXor2>((int)v));
    v = min(v, (unsigned)dpp<kRowHalfMirror>((int)v));
    return v;
}

// One search level with step S.  wrow0 / wcol0: window coordinates of the macroblock origin
// row (already + lane row i) and column.  Candidate (my, mx) sits at displacement
// (st.cx + mx*S, st.cy + my*S).  FIRST: the centre's error is not known yet (first level).
// BOUNDS: check every candidate against the plane (src/common.rs:171, :182); false for wavefronts whose whole
// strip lies at least 15 pixels inside the plane, where no candidate can leave it.
// Per candidate the lane accumulates  sum b^2 - 2 sum ab  over its two rows.  In the aligned levels (S = 8, 4) the three
// horizontal candidates of one row read overlapping dwords of the same span, so their sums of squares share the dwords
// they have in common (16 instead of 24 v_dot4 per row pair at S = 8, 12 instead of 24 at S = 4).
// DPPRED: the 8 partials reach the lane that owns the candidate by a three-step reduce-scatter in registers (pairings i ^ 7, i ^ 2, i ^ 1:
// each step a lane keeps the half of its values whose owners lie on its side and adds the partner's partials of those; 14 selects + 7 DPP
// adds) instead of the LDS transpose -- no reduction region, which is what lets k_pf_search hold more than six workgroups per CU.
template <int S, bool FIRST, bool BOUNDS, bool DPPRED = false>
__device__ __forceinline__ void search_level(const uint8_t *win, int wrow0, int wcol0, const uint4 &top, const uint4 &bot,
                                             int a2, int mbx, int mby, int pw, int ph, SearchState &st, const SearchLane &sl)
{
    constexpr bool kAligned = (S >= 4);            // displacement is a multiple of 4 pixels: no byte shifts
    const int sh = (st.cx - 1) & 3;                // only used at step 1: phase of the leftmost candidate
    // span origin (bytes from the window row start): dword aligned
    const int col = kAligned ? (wcol0 + st.cx - S) : (S == 2 ? wcol0 + st.cx - 4 : wcol0 + ((st.cx - 1) & ~3));
    int part[3][3];
    if (S == 8) {
        // Step 8 (always the first level): the candidate rows are 8 apart, so the bottom row of candidate row my is the top row of
        // my + 1 -- four distinct reference rows per lane instead of six; each is read once, its three window sums of squares are
        // computed once (8 v_dot4 + 1 add) and serve both candidate rows that contain it.
        unsigned ab[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, bb[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint8_t *rp = win + (wrow0 + st.cy + (j - 1) * 8) * kWinStride + col;   // col = 16m + 8 (mod 16 == 8): 8-, 16-, 8-byte pieces
            const uint2 p0 = *reinterpret_cast<const uint2 *>(rp), p2 = *reinterpret_cast<const uint2 *>(rp + 24);
            const uint4 p1 = *reinterpret_cast<const uint4 *>(rp + 8);
            const unsigned d[8] = {p0.x, p0.y, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y};
            const unsigned q23 = sq2(d[2], d[3], 0), q45 = sq2(d[4], d[5], 0);
            const unsigned rsq[3] = {sq2(d[0], d[1], q23), q23 + q45, sq2(d[6], d[7], q45)};
#pragma unroll
            for (int c = 0; c < 3; c++) {   // candidates at dwords 0, 2, 4
                if (j <= 2) {
                    ab[j][c] = dot_ab(top, d[2 * c], d[2 * c + 1], d[2 * c + 2], d[2 * c + 3], ab[j][c]);
                    bb[j][c] += rsq[c];
                }
                if (j >= 1) {
                    ab[j - 1][c] = dot_ab(bot, d[2 * c], d[2 * c + 1], d[2 * c + 2], d[2 * c + 3], ab[j - 1][c]);
                    bb[j - 1][c] += rsq[c];
                }
            }
        }
#pragma unroll
        for (int my = 0; my < 3; my++)
#pragma unroll
            for (int c = 0; c < 3; c++) part[my][c] = __mul24((int)ab[my][c], sl.neg2) + (int)bb[my][c];
    } else
#pragma unroll
    for (int my = -1; my <= 1; my++) {
        const uint8_t *rp = win + (wrow0 + st.cy + my * S) * kWinStride + col;
        const bool centre_known = !FIRST && my == 0;   // (0,0) is not evaluated again (:176)
        unsigned ab[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
        if (S == 4) {   // dword, 16-byte, dword; candidates at dwords 0, 1, 2 of a 6-dword span
            const unsigned *t = reinterpret_cast<const unsigned *>(rp), *b = reinterpret_cast<const unsigned *>(rp + 8 * kWinStride);
            uint4 t1, b1;     // col + 4 is a multiple of 8 only: two 8-byte reads each (no 16-byte access at an 8-byte aligned address)
            {
                const uint2 ta = *reinterpret_cast<const uint2 *>(rp + 4), tb = *reinterpret_cast<const uint2 *>(rp + 12);
                const uint2 ba = *reinterpret_cast<const uint2 *>(rp + 8 * kWinStride + 4), bc = *reinterpret_cast<const uint2 *>(rp + 8 * kWinStride + 12);
                t1 = make_uint4(ta.x, ta.y, tb.x, tb.y);
                b1 = make_uint4(ba.x, ba.y, bc.x, bc.y);
            }
            const unsigned dT[6] = {t[0], t1.x, t1.y, t1.z, t1.w, t[5]};
            const unsigned dB[6] = {b[0], b1.x, b1.y, b1.z, b1.w, b[5]};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                ab[c] = dot_ab(bot, dB[c], dB[c + 1], dB[c + 2], dB[c + 3], dot_ab(top, dT[c], dT[c + 1], dT[c + 2], dT[c + 3], 0));
            }
            const unsigned q23 = sq2(dB[2], dB[3], sq2(dT[2], dT[3], 0));                 // common to all three windows
            if (centre_known) {
                bb[0] = sq2(dB[0], dB[1], sq2(dT[0], dT[1], q23));
                bb[2] = sq2(dB[4], dB[5], sq2(dT[4], dT[5], q23));
            } else {
                const unsigned y = __builtin_amdgcn_udot4(dB[1], dB[1], __builtin_amdgcn_udot4(dT[1], dT[1], q23, false), false);   // q23 + s1
                const unsigned z = __builtin_amdgcn_udot4(dB[4], dB[4], __builtin_amdgcn_udot4(dT[4], dT[4], 0, false), false);     // s4
                bb[0] = __builtin_amdgcn_udot4(dB[0], dB[0], __builtin_amdgcn_udot4(dT[0], dT[0], y, false), false);
                bb[1] = y + z;
                bb[2] = __builtin_amdgcn_udot4(dB[5], dB[5], __builtin_amdgcn_udot4(dT[5], dT[5], q23 + z, false), false);
            }
        } else if (S == 2) {   // steps 8 and 4 came first: st.cx is a multiple of 4, the candidates sit 2 bytes before, on and 2 bytes after a dword boundary
            const unsigned *t = reinterpret_cast<const unsigned *>(rp), *b = reinterpret_cast<const unsigned *>(rp + 8 * kWinStride);
            unsigned rt[6], rb[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { rt[k] = t[k]; rb[k] = b[k]; }      // bytes cx - 4 .. cx + 19
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                unsigned eT[4], eB[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int o = k + (c >> 1);
                    eT[k] = c == 1 ? rt[k + 1] : __builtin_amdgcn_alignbyte(rt[o + 1], rt[o], 2);
                    eB[k] = c == 1 ? rb[k + 1] : __builtin_amdgcn_alignbyte(rb[o + 1], rb[o], 2);
                }
                ab[c] = dot_ab(bot, eB[0], eB[1], eB[2], eB[3], dot_ab(top, eT[0], eT[1], eT[2], eT[3], 0));
#ifdef PFV_ABL_SEARCH_BB   // ablation (results invalid): what a sliding sum of squares could save at most -- the off-centre rows' squares cost nothing
                if (my != 0) continue;
#endif
                bb[c] = sq4(eB[0], eB[1], eB[2], eB[3], sq4(eT[0], eT[1], eT[2], eT[3], 0));
            }
        } else {
            const unsigned *t = reinterpret_cast<const unsigned *>(rp), *b = reinterpret_cast<const unsigned *>(rp + 8 * kWinStride);
            static_assert(S == 1 || S == 2 || S >= 4, "the byte-granular path is the 1-pixel level");
            // step 1: candidates at bytes cx - 1, cx, cx + 1.  Rebase once to the leftmost one (lane-variable shift), then
            // compile-time shifts of 0, 1, 2 bytes
            unsigned rt[6], rb[6], dT[5], dB[5];
#pragma unroll
            for (int k = 0; k < 6; k++) { rt[k] = t[k]; rb[k] = b[k]; }
#pragma unroll
            for (int k = 0; k < 5; k++) {   // e[k] = bytes starting at (cx - 1) + 4k
                dT[k] = __builtin_amdgcn_alignbyte(rt[k + 1], rt[k], sh);
                dB[k] = __builtin_amdgcn_alignbyte(rb[k + 1], rb[k], sh);
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                unsigned eT[4], eB[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    eT[k] = c ? __builtin_amdgcn_alignbyte(dT[k + 1], dT[k], c) : dT[k];
                    eB[k] = c ? __builtin_amdgcn_alignbyte(dB[k + 1], dB[k], c) : dB[k];
                }
                ab[c] = dot_ab(bot, eB[0], eB[1], eB[2], eB[3], dot_ab(top, eT[0], eT[1], eT[2], eT[3], 0));
#ifdef PFV_ABL_SEARCH_BB
                if (my != 0) continue;
#endif
                bb[c] = sq4(eB[0], eB[1], eB[2], eB[3], sq4(eT[0], eT[1], eT[2], eT[3], 0));
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) part[my + 1][c] = __mul24((int)ab[c], sl.neg2) + (int)bb[c];   // sum ab < 2^22
    }
    int err;
    if (DPPRED) {
        // v[c]: this lane's partial of candidate c (visiting order); lane i ends up with the total of candidate i
        const int v[8] = {part[0][0], part[0][1], part[0][2], part[1][0], part[1][2], part[2][0], part[2][1], part[2][2]};
        const int i = sl.ord - 1;
        const bool b2 = (i & 4) != 0, b1 = (i & 2) != 0, b0 = (i & 1) != 0;
        int k4[4], k2[2];
#pragma unroll
        for (int k = 0; k < 4; k++) k4[k] = (b2 ? v[4 + k] : v[k]) + dpp<kRowHalfMirror>(b2 ? v[k] : v[4 + k]);
#pragma unroll
        for (int k = 0; k < 2; k++) k2[k] = (b1 ? k4[2 + k] : k4[k]) + dpp<kQuadXor2>(b1 ? k4[k] : k4[2 + k]);
        err = a2 + (b0 ? k2[1] : k2[0]) + dpp<kQuadXor1>(b0 ? k2[0] : k2[1]);
    } else {
    // transposed reduction: lane c of the macroblock collects the 8 row-pair partials of candidate c
    int ord = 0;
#pragma unroll
    for (int my = -1; my <= 1; my++) {
#pragma unroll
        for (int mx = -1; mx <= 1; mx++) {
            if (my == 0 && mx == 0) continue;
            sl.wr[ord * 8] = part[my + 1][mx + 1];
            ord++;
        }
    }
    wave_lds_sync();
    const int4 x = sl.rd[0], y = sl.rd[1];
    wave_lds_sync();
    err = a2 + ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
    }
    // this lane's candidate against the plane, then the sequential accept rule of the reference: strict `<`, first
    // visited wins (:189), centre first -- the lexicographic minimum of (error, visiting order)
    bool valid = true;
    if (BOUNDS) {
        const int oy = mby + st.cy + sl.dy * S, ox = mbx + st.cx + sl.dx * S;
        valid = oy >= 0 && oy <= ph - 16 && ox >= 0 && ox <= pw - 16;        // :171, :182
    }
    const unsigned key = valid ? (((unsigned)err << 4) | (unsigned)sl.ord) : 0xffffffffu;
    const unsigned centre = FIRST ? ((unsigned)(a2 + mb_sum(part[1][1])) << 4) : ((unsigned)st.err << 4);
    const unsigned best = min(mb_min(key), centre);
    const int bo = (int)(best & 15u);
    st.err = (int)(best >> 4);
    // winner -> step: two-bit fields indexed by the visiting order (0 = the centre stays): my + 1 and mx + 1
    st.cy += ((int)((0x2A501u >> (2 * bo)) & 3u) - 1) * S;
    st.cx += ((int)((0x24891u >> (2 * bo)) & 3u) - 1) * S;
}

// Geometry of one p-frame tile (128 x 64 px = 4 vertically stacked strips) as seen by one wavefront.
#ifndef PFV_PENC_WAVES
#define PFV_PENC_WAVES 5   // wavefronts per SIMD the p-frame encoder is compiled for: 94 VGPRs, 27 KiB of LDS per workgroup.  Six was tried in round 3 (quantiser
                           // table moved into the wavefronts' reduction regions -> 26 KiB; 80 VGPRs cost 16 spills): 619 vs 585 us on one box
#endif
struct TilePos {
    StripPos sp;        // this wavefront's strip
    int plane_ty;       // tile row inside the plane
    int winx0, winy0;   // plane coordinates of the window origin
    bool wave_valid;    // the strip exists (the last tile row of a plane may be partial)
};
__device__ __forceinline__ TilePos locate_tile(const FrameGeom &g, int vt, int wave)
{
    TilePos t;
    t.sp.stream = vt / g.tiles_per_frame;
    int tl = vt - t.sp.stream * g.tiles_per_frame;
    t.sp.plane = plane_of(g, tl, true);
    const PlaneGeom &p = g.p[t.sp.plane];
    tl -= p.tile0;
    t.plane_ty = tl / p.strips_x;
    t.sp.sx = tl - t.plane_ty * p.strips_x;
    t.sp.by = t.plane_ty * kStripsPerWG + wave;
    finish_strip(g, t.sp);
    t.wave_valid = t.sp.by < p.bh;
    t.winx0 = t.sp.x0 - 16;
    t.winy0 = t.plane_ty * (16 * kStripsPerWG) - 15;
    return t;
}

// Stage the tile's reference window with direct global->LDS loads (global_load_lds_dwordx4: no VGPR
// round trip, completion tracked by vmcnt and drained at the next workgroup barrier).  The LDS image of one
// wave-instruction is lane-linear (base + lane * 16), so a window row is 11 chunks of 16 B (160 B of pixels +
// 16 B pad = kWinStride); wavefront w issues a contiguous run of 5 (w = 0) or 4 of the 17 1-KiB loads and later owns
// that slice as its exchange region.  Chunks outside the plane (or past the last window row) are redirected to a clamped in-plane
// address so that every lane stays active (the LDS address is derived from the first active lane); their
// content is never used (candidates outside the plane are invalid).
__device__ __forceinline__ void issue_window(const PlaneGeom &p, const uint8_t *refp, const TilePos &t, uint8_t *winbuf, int wave,
                                             int lane)
{
    const int q0 = win_first_issue(wave), q1 = win_first_issue(wave + 1);
#pragma unroll 5
    for (int q = q0; q < q1; q++) {
        const int c = q * 64 + lane;
        const int row = c / kWinChunksPerRow, col = c - row * kWinChunksPerRow;
        const int y = min(max(t.winy0 + row, 0), p.ph - 1), x = min(max(t.winx0 + col * 16, 0), p.pw - 16);
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(refp + (long)y * p.pw + x), (lds_void_t *)(winbuf + c * 16), 16, 0, 0);
    }
}

// Result of the search phase of one macroblock (identical in its 8 lanes) plus the lane's patch rows.
struct SearchOut {
    int cx, cy;
    bool coded;
    uint4 patch[2];
};

// Phase 1 of a tile for one wavefront: motion search, skip decision, fetch of the chosen patch rows.
// Reads the window; issues no global memory operation.
template <bool DPPRED = false>
__device__ __forceinline__ void penc_search(const FrameGeom &g, const TilePos &tp, const uint8_t *win, int *red, const uint4 (&rows)[2], int lane,
                                            float min_err, int neg2, SearchOut &so)
{
    const StripPos &sp = tp.sp;
    const PlaneGeom &p = g.p[sp.plane];
    const int m = lane >> 3, i = lane & 7;
    const bool mb_valid = m < sp.n_mb;
    const int mbx = sp.x0 + m * 16, mby = sp.y0;
    const int wcol0 = mbx - tp.winx0;              // = 16 + 16 m
    const int wrow0 = mby - tp.winy0 + i;          // window row of source row i

    // sum of squares of the source block (2 rows per lane)
    unsigned s2 = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        s2 = __builtin_amdgcn_udot4(rows[h].x, rows[h].x, s2, false);
        s2 = __builtin_amdgcn_udot4(rows[h].y, rows[h].y, s2, false);
        s2 = __builtin_amdgcn_udot4(rows[h].z, rows[h].z, s2, false);
        s2 = __builtin_amdgcn_udot4(rows[h].w, rows[h].w, s2, false);
    }
    const int a2 = mb_sum((int)s2);

    // 4-step search (reference src/common.rs:154-204, steps 8, 4, 2, 1)
    SearchState st;
    st.cx = 0; st.cy = 0; st.err = 0;
    const SearchLane sl = make_search_lane(red, m, i, neg2);
    // strips at least 15 px inside the plane on every side (84 % of a 1080p luma plane) skip the bounds tests
    const bool interior = sp.x0 >= 16 && sp.x0 + kStripMB * 16 + 16 <= p.pw && sp.y0 >= 16 && sp.y0 + 32 <= p.ph;   // wave-uniform
    if (interior) {
        search_level<8, true, false, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        KMARK(3);
        search_level<4, false, false, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        KMARK(4);
        search_level<2, false, false, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        KMARK(5);
        search_level<1, false, false, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        KMARK(6);
    } else {
        search_level<8, true, true, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        search_level<4, false, true, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        search_level<2, false, true, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
        search_level<1, false, true, DPPRED>(win, wrow0, wcol0, rows[0], rows[1], a2, mbx, mby, p.pw, p.ph, st, sl);
    }

    // skip decision (src/common.rs:209, :221): best_err <= px_err^2 * 256, compared in f32
    so.coded = mb_valid && !((float)st.err <= min_err);
    so.cx = st.cx; so.cy = st.cy;

    // the lane's two rows of the chosen patch (get_block of the reconstruction, :261)
    const int wx = wcol0 + st.cx, shp = wx & 3;
    const uint8_t *rp = win + (wrow0 + st.cy) * kWinStride + (wx & ~3);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned *d = reinterpret_cast<const unsigned *>(rp + 8 * h * kWinStride);
        unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
        so.patch[h] = make_uint4(__builtin_amdgcn_alignbyte(d1, d0, shp), __builtin_amdgcn_alignbyte(d2, d1, shp),
                                 __builtin_amdgcn_alignbyte(d3, d2, shp), __builtin_amdgcn_alignbyte(d4, d3, shp));
    }
}

// One half (h: rows 0..7 or 8..15 = two subblocks per lane) of the residual pipeline of a wavefront's 8 macroblock slots:
// calc_residuals -> encode_subblock_delta -> [store(): the zigzag stage xw now holds the quantised coefficients of the 8 slots]
// -> decode_subblock -> apply_residuals (src/common.rs:108-123, 300-311, 313-325, 98-104).  Returns the lane's reconstructed
// 16-pixel row.  A slot whose source row equals its prediction row (skipped or empty slot) yields zero coefficients and the
// prediction itself.
// REUNPACK (k_pf_transform): the prediction row stays PACKED (4 registers) across the transforms and is converted to floats a second time
// for the reconstruction, instead of 16 float registers living through both transforms -- 16 conversions per half for 12 registers.
__device__ __forceinline__ void opaque_row(uint4 &r)   // the compiler must not know this is the row it has already unpacked
{
#ifndef PFV_HIPEMU
    asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
#else
    (void)r;
#endif
}
template <bool FLT, bool REUNPACK = false, class Store>
__device__ __forceinline__ uint4 penc_half(const uint4 &srow, const uint4 &prow, int *xw, int m, int i, const LaneQ &lq, float qmagic, bool want_recon,
                                           Store &&store)
{
    uint4 out = prow;
    if (FLT) {
        f2 x[8], pp[8];
        unpack_row_f(srow, x);
        unpack_row_f(prow, pp);
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = residual_f(x[k], pp[k]);   // calc_residuals (:118-119), delta / 2 truncating, << 8 (:304)
        forward_half_f(x, xw, m, i, lq, qmagic);
        store();
        wave_lds_sync();
        if (want_recon) {
            inverse_half_f<false>(x, xw, m, i, lq);
            if (REUNPACK) {
                uint4 again = prow;
                opaque_row(again);
                unpack_row_f(again, pp);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {   // apply_residuals (:98-104): prev + 2 * min(t, 127), saturated by the pack
                const f2 t = f2{__builtin_fminf(x[k][0], 127.0f), __builtin_fminf(x[k][1], 127.0f)};
                pp[k] = pp[k] + t * f2s(2.0f);
            }
            out = pack_row_f(pp);
        }
    } else {
        int v[2][8], pp[2][8];
        unpack_row(srow, v);
        unpack_row(prow, pp);
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int d = v[s][k] - pp[s][k];                       // calc_residuals (:118-119); |d| <= 255
                v[s][k] = (int)((unsigned)tdiv2(d) << 8);         // (:304)
            }
        }
        forward_half(v, xw, m, i, lq, true);
        store();
        wave_lds_sync();
        if (want_recon) {
            inverse_half(v, xw, m, i, lq);
#pragma unroll
            for (int s = 0; s < 2; s++) {
#pragma unroll
                for (int k = 0; k < 8; k++)   // apply_residuals (:98-104); v == 0 for skipped blocks: copy (:281-283)
                    pp[s][k] = min(max(pp[s][k] + 2 * v[s][k], 0), 255);
            }
            out = pack_row(pp);
        }
    }
    return out;
}

// What a skipped macroblock leaves behind (src/common.rs:221-222 returns subblocks: None, :281-283 copies the patch): zero
// coefficients (API contract: fixed 512-byte stride) and the prediction as reconstruction.  Called by the 8 lanes of the macroblock.
__device__ __forceinline__ void penc_store_skipped(int16_t *coef_mb, uint8_t *dst, long pw, int i, const uint4 (&patch)[2])
{
#pragma unroll
    for (int j = 0; j < 4; j++) st_stream(&reinterpret_cast<uint4 *>(coef_mb)[j * 8 + i], make_uint4(0, 0, 0, 0));
    if (dst) {
        *reinterpret_cast<uint4 *>(dst) = patch[0];
        *reinterpret_cast<uint4 *>(dst + 8 * pw) = patch[1];
    }
}

// Phase 2, strip form: the wavefront transforms its own 8 macroblocks (all of them when any is coded: a skipped one takes its
// prediction as its source).  xw: this wavefront's exchange region (its own part of the window buffer that has just been released).
template <bool FLT>
__device__ __forceinline__ void penc_transform(const FrameGeom &g, const TilePos &tp, const SearchOut &so, const uint4 (&rows)[2], int *xw,
                                               int lane, int16_t *__restrict__ coef, uint8_t *__restrict__ recon, const int *qtab_lds, float qmagic)
{
    const StripPos &sp = tp.sp;
    const PlaneGeom &p = g.p[sp.plane];
    const int m = lane >> 3, i = lane & 7;
    const bool mb_valid = m < sp.n_mb, coded = so.coded;
    const int mbx = sp.x0 + m * 16;
    int16_t *coef_mb0 = coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256;
    uint8_t *dst = recon ? recon + (long)sp.stream * g.pad_frame_bytes + p.pad_off + (long)(sp.y0 + i) * p.pw + mbx : nullptr;

    if (__any(coded)) {   // wavefront-uniform: the LDS transposes need all lanes
        const LaneQ lq{qtab_lds, i};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // a skipped macroblock (None in the reference, zero coefficients here) takes its prediction as its source:
            // zero residual -> zero coefficients -> reconstruction = prediction, without masking 16 values per pass
            const uint4 o = penc_half<FLT>(coded ? rows[h] : so.patch[h], so.patch[h], xw, m, i, lq, qmagic, recon != nullptr,
                                           [&]() { store_coef_half(xw, coef_mb0, sp.n_mb, lane, h); });
            if (h == 0) KMARK(9);
            if (recon && mb_valid) *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = o;
        }
    } else if (mb_valid) {
        penc_store_skipped(coef_mb0 + m * 256, dst, p.pw, i, so.patch);   // every macroblock of the strip is skipped
    }
}

// Phase 2, compacted form (skip-aware transform): the tile's CODED macroblocks have been renumbered 0 .. n_coded-1 in tile
// order and their source / prediction rows staged in LDS; this wavefront transforms slots 8 * wave .. 8 * wave + 7 and stores the
// results at the macroblocks' own places.  slot_orig: LDS, per slot of the TILE the macroblock's place (strip << 3 | macroblock
// in the strip).  Lanes of an empty slot (slot >= n_coded) carry zeros and store nothing.
template <bool FLT>
__device__ __forceinline__ void penc_transform_compact(const FrameGeom &g, const TilePos &tp, int wave, int n_coded, const int *slot_orig,
                                                       const uint4 *staged, int *xw, int lane, int16_t *__restrict__ coef,
                                                       uint8_t *__restrict__ recon, const int *qtab_lds, float qmagic)
{
    const PlaneGeom &p = g.p[tp.sp.plane];
    const int m = lane >> 3, i = lane & 7;
    const int slot = wave * 8 + m;
    const bool has = slot < n_coded;
    // the tile's first strip starts at macroblock row plane_ty * kStripsPerWG; strip s of the tile is s macroblock rows further down
    const long tile_mb0 = (long)tp.sp.stream * g.mbs_per_frame + p.mb0 + (long)(tp.plane_ty * kStripsPerWG) * p.bw + tp.sp.sx * kStripMB;
    const LaneQ lq{qtab_lds, i};
    const uint4 *sl = staged + (has ? slot : 0) * 32 + i * 4;   // the lane's 64 staged bytes: source rows 0 / 8, prediction rows 0 / 8
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // the rows are fetched half by half (the staging area lives until the tile is done): 8 registers live instead of 16
        uint4 srow = sl[h], prow = sl[2 + h];
        if (!has) srow = prow = make_uint4(0, 0, 0, 0);
        const uint4 o = penc_half<FLT>(srow, prow, xw, m, i, lq, qmagic, recon != nullptr, [&]() {
#pragma unroll
            for (int j = 0; j < 2; j++) {   // the stage's 128 16-byte chunks, 16 per slot, to the slots' macroblocks
                const int ch = j * 64 + lane, sc = wave * 8 + (ch >> 4);
                if (sc < n_coded) {
                    const int og = slot_orig[sc];
                    int16_t *mb = coef + (tile_mb0 + (long)(og >> 3) * p.bw + (og & 7)) * 256;
                    st_stream(&reinterpret_cast<uint4 *>(mb)[h * 16 + (ch & 15)], reinterpret_cast<const uint4 *>(xw)[stage_chunk(ch)]);
                }
            }
        });
        if (recon && has) {
            // the macroblock's place is looked up HERE, after the pipeline (an LDS read cannot move up across its hand-offs): the
            // row pointer is not alive during the transforms, which need every register they can get
            const int orig = slot_orig[slot];
            uint8_t *dst = recon + (long)tp.sp.stream * g.pad_frame_bytes + p.pad_off +
                           (long)((tp.plane_ty * kStripsPerWG + (orig >> 3)) * 16 + i + 8 * h) * p.pw + tp.sp.x0 + (orig & 7) * 16;
            *reinterpret_cast<uint4 *>(dst) = o;
        }
    }
}

// One workgroup per tile (128 x 64 px = 4 vertically stacked strips).  Per wavefront:
//     LDS-DMA of the tile's window (each wavefront its own 5 KiB slice), source rows -> registers
//     ---- workgroup barrier: window complete ----
//     search (partial sums transposed through the wavefront's reduction region), fetch the chosen patch rows
//     ---- workgroup barrier: window released ----
//     transform + reconstruct + store; the exchange region lives in the wavefront's own window slice
// LDS per workgroup: 17 KiB window (+ 16 bytes in front of it: the 1-pixel level reads one dword to the left of the
// leftmost candidate of the first window row) + 9 KiB reduction regions + 1 KiB quantiser tables.
template <bool FLT>
__global__ __launch_bounds__(kThreads) PFV_WAVES_PER_EU(PFV_PENC_WAVES) void k_enc_pframe(FrameGeom g, const uint8_t *__restrict__ src,
                                                          const uint8_t *__restrict__ ref, int8_t *__restrict__ mv_out,
                                                          uint8_t *__restrict__ has_out, int16_t *__restrict__ coef,
                                                          uint8_t *__restrict__ recon, const QTab *__restrict__ qtabs,
                                                          float min_err, int neg2, float qmagic, int compact_max)
{
    __shared__ __attribute__((aligned(16))) uint8_t win_lds[16 + kWinAlloc];
    __shared__ __attribute__((aligned(16))) int red_lds[kStripsPerWG][kRedDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kQTabDwords];
    uint8_t *win = win_lds + 16;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPRs
    const int m = lane >> 3, i = lane & 7;
    const int vt = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const TilePos cur = locate_tile(g, vt, wave);
    const PlaneGeom &p = g.p[cur.sp.plane];
    if (wave == 0) fill_qtable<true, FLT>(qtab_lds, qtabs + p.qsel, lane);   // one copy per workgroup (a tile lies in one plane)

    KMARK(0);
    KMARK_WHERE();
    issue_window(p, ref + (long)cur.sp.stream * g.pad_frame_bytes + p.pad_off, cur, win, wave, lane);
    uint4 rows[2];
    rows[0] = rows[1] = make_uint4(0, 0, 0, 0);
    if (cur.wave_valid) {
        const uint8_t *plane = frame_src(g, src, cur.sp.stream) + p.src_off;
        rows[0] = load_src16(plane, p, cur.sp.x0 + m * 16, cur.sp.y0 + i);
        rows[1] = load_src16(plane, p, cur.sp.x0 + m * 16, cur.sp.y0 + i + 8);
    }
    KMARK(1);
    __syncthreads();   // window complete (vmcnt drained at the barrier)
    KMARK(2);
    SearchOut so;
    so.cx = so.cy = 0; so.coded = false;
    so.patch[0] = so.patch[1] = make_uint4(0, 0, 0, 0);
    if (cur.wave_valid) penc_search(g, cur, win, red_lds[wave], rows, lane, min_err, neg2, so);
    KMARK(7);
    const bool mb_valid = cur.wave_valid && m < cur.sp.n_mb;
    if (i == 0 && mb_valid) {   // block headers (DeltaEncodedMacroBlock.motion_x / _y, subblocks.is_some(); src/common.rs:14-19)
        const long mbi = (long)cur.sp.stream * g.mbs_per_frame + cur.sp.mb_first + m;
        mv_out[mbi * 2 + 0] = (int8_t)so.cx;
        mv_out[mbi * 2 + 1] = (int8_t)so.cy;
        has_out[mbi] = so.coded ? 1 : 0;
    }
    // Which macroblocks of the TILE are coded: one bit per macroblock, a byte per strip, published in the last 8 dwords of the
    // last reduction region (no search touches them: kRedPitch * kStripMB - 8 = 568 is past the last partial sum).
    constexpr int kMaskAt = (kStripsPerWG - 1) * kRedDwords + kRedDwords - 8;
    int *red_all = &red_lds[0][0];
    {
        const unsigned long long bal = __ballot(so.coded);
        unsigned mine = 0;
#pragma unroll
        for (int k = 0; k < kStripMB; k++) mine |= (unsigned)((bal >> (8 * k)) & 1ull) << k;
        if (lane == 0) red_all[kMaskAt + wave] = (int)mine;
    }
    __syncthreads();   // window released by every wavefront; the strips' masks are visible
    KMARK(8);
    unsigned tile_mask = 0;
#pragma unroll
    for (int k = 0; k < kStripsPerWG; k++) tile_mask |= (unsigned)__builtin_amdgcn_readfirstlane(red_all[kMaskAt + k]) << (8 * k);
    int n_coded = __builtin_popcount(tile_mask), strips_coded = 0;
#pragma unroll
    for (int k = 0; k < kStripsPerWG; k++) strips_coded += ((tile_mask >> (8 * k)) & 0xffu) ? 1 : 0;
    int *xw = reinterpret_cast<int *>(win + win_first_issue(wave) * 1024);
    // Skip-aware transform.  The reference transforms nothing for a skipped macroblock (src/common.rs:221-222); a wavefront,
    // however, runs the transform for all 8 of its macroblocks as soon as one of them is coded.  When the tile's coded
    // macroblocks fit fewer wavefronts than the strips that hold them (and at most kCompactMax, which is what the staging
    // area takes), they are moved together: rows through LDS, one more barrier, and only ceil(n / 8) wavefronts transform.
    // compact_max: kCompactMax, or 0 to switch the compaction off (pfv_ctx_set_option(PFV_OPT_TILE_COMPACTION, 0): measurements).
    constexpr int kCompactMax = kPencCompactMax;          // 16 slots x 512 B = 8 KiB of the 9 KiB of reduction regions
    constexpr int kSlotOrigAt = kCompactMax * 128;        // dwords; the slots' places follow the staging area
    static_assert(kSlotOrigAt + kCompactMax <= kMaskAt, "staging area + slot table must fit below the strip masks");
    static_assert(kStripsPerWG * 8 <= 32, "one bit per macroblock of the tile");
    if (n_coded <= compact_max && ((n_coded + 7) >> 3) < strips_coded) {   // workgroup-uniform
        uint4 *stage4 = reinterpret_cast<uint4 *>(red_all);
        const int place = wave * 8 + m;
        if (so.coded) {
            const int slot = __builtin_popcount(tile_mask & ((1u << place) - 1u));
            uint4 *sl = stage4 + slot * 32 + i * 4;         // 64 bytes per lane: source rows, prediction rows
            sl[0] = rows[0]; sl[1] = rows[1]; sl[2] = so.patch[0]; sl[3] = so.patch[1];
            if (i == 0) red_all[kSlotOrigAt + slot] = place;
        } else if (mb_valid) {
            const PlaneGeom &pp = g.p[cur.sp.plane];
            penc_store_skipped(coef + ((long)cur.sp.stream * g.mbs_per_frame + cur.sp.mb_first + m) * 256,
                               recon ? recon + (long)cur.sp.stream * g.pad_frame_bytes + pp.pad_off + (long)(cur.sp.y0 + i) * pp.pw + cur.sp.x0 + m * 16 : nullptr,
                               pp.pw, i, so.patch);
        }
        __syncthreads();   // rows staged
        if (wave * 8 < n_coded)
            penc_transform_compact<FLT>(g, cur, wave, n_coded, red_all + kSlotOrigAt, stage4, xw, lane, coef, recon, qtab_lds, qmagic);
    } else if (cur.wave_valid) {
        penc_transform<FLT>(g, cur, so, rows, xw, lane, coef, recon, qtab_lds, qmagic);
    }
    KMARK(11);
}

// ================================================================== P-frame encode, SPLIT form (round 6): search kernel + transform kernel
// The same operator as k_enc_pframe (encode_plane_delta + the closed loop's decode_plane_delta, src/common.rs:388-421, 448-475) as two
// launches, so that each phase runs at the occupancy ITS registers allow and the transform runs on coded macroblocks only:
//   k_pf_search     one workgroup per 128 x 64 tile as before: window DMA, 4-step search (src/common.rs:154-204), skip decision (:209, :221)
//                   -> motion vectors and flags; a SKIPPED macroblock is finished here (zero coefficients at the fixed 512-byte stride, the
//                   prediction as its reconstruction, :281-283: its patch is in the LDS window anyway); a coded one is left to the second
//                   kernel.  No transform registers, no quantiser table, no window-release barrier.
//   k_pf_transform  one wavefront per GROUP of kTfStrips consecutive strips of one plane of one stream (64 macroblocks): reads the group's
//                   flags, numbers its coded macroblocks 0 .. n-1 (ballot + mbcnt, table in LDS) and transforms them 8 per pass: source rows,
//                   patch rows by motion vector from the previous reconstruction (get_block, :327-339, as k_dec_pframe fetches them),
//                   residual -> forward DCT -> quantise -> store -> closed-loop inverse -> reconstruction (penc_half).  ceil(n / 8) passes
//                   instead of one pass per strip that holds a coded macroblock: no work on skipped slots beyond the last pass's tail.
// Extra traffic against the fused kernel: a coded macroblock's 256 source bytes and its 256-byte patch are read a second time.
template <bool DPPRED>
__global__ __launch_bounds__(kThreads) PFV_WAVES_PER_EU(kPsWaves) void k_pf_search(FrameGeom g, const uint8_t *__restrict__ src,
                                                          const uint8_t *__restrict__ ref, int8_t *__restrict__ mv_out,
                                                          uint8_t *__restrict__ has_out, int16_t *__restrict__ coef,
                                                          uint8_t *__restrict__ recon, float min_err, int neg2)
{
    __shared__ __attribute__((aligned(16))) uint8_t win_lds[16 + kWinAlloc];
    __shared__ __attribute__((aligned(16))) int red_lds[DPPRED ? 1 : kStripsPerWG][DPPRED ? 4 : kRedDwords];
    uint8_t *win = win_lds + 16;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = lane >> 3, i = lane & 7;
    const int vt = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const TilePos cur = locate_tile(g, vt, wave);
    const PlaneGeom &p = g.p[cur.sp.plane];
    issue_window(p, ref + (long)cur.sp.stream * g.pad_frame_bytes + p.pad_off, cur, win, wave, lane);
    uint4 rows[2];
    rows[0] = rows[1] = make_uint4(0, 0, 0, 0);
    if (cur.wave_valid) {
        const uint8_t *plane = frame_src(g, src, cur.sp.stream) + p.src_off;
        rows[0] = load_src16(plane, p, cur.sp.x0 + m * 16, cur.sp.y0 + i);
        rows[1] = load_src16(plane, p, cur.sp.x0 + m * 16, cur.sp.y0 + i + 8);
    }
    __syncthreads();   // window complete (vmcnt drained at the barrier); the only barrier of the kernel
    if (!cur.wave_valid) return;
    SearchOut so;
    penc_search<DPPRED>(g, cur, win, red_lds[DPPRED ? 0 : wave], rows, lane, min_err, neg2, so);
    if (m < cur.sp.n_mb) {
        const long mbi = (long)cur.sp.stream * g.mbs_per_frame + cur.sp.mb_first + m;
        if (i == 0) {   // block headers (DeltaEncodedMacroBlock.motion_x / _y, subblocks.is_some(); src/common.rs:14-19)
            mv_out[mbi * 2 + 0] = (int8_t)so.cx;
            mv_out[mbi * 2 + 1] = (int8_t)so.cy;
            has_out[mbi] = so.coded ? 1 : 0;
        }
        if (!so.coded)
            penc_store_skipped(coef + mbi * 256, recon ? recon + (long)cur.sp.stream * g.pad_frame_bytes + p.pad_off + (long)(cur.sp.y0 + i) * p.pw + cur.sp.x0 + m * 16 : nullptr,
                               p.pw, i, so.patch);
    }
}

constexpr int kTfStrips = 8;                       // strips per wavefront of k_pf_transform: 64 macroblocks = one flag per lane
// groups of one plane of one frame
__host__ __device__ __forceinline__ int tf_groups_of_plane(const PlaneGeom &p) { return (p.strips_x * p.bh + kTfStrips - 1) / kTfStrips; }
__host__ __device__ __forceinline__ int tf_groups_per_frame(const FrameGeom &g)
{
    int n = 0;
    for (int k = 0; k < g.n_planes; k++) n += tf_groups_of_plane(g.p[k]);
    return n;
}

template <bool FLT>
__global__ __launch_bounds__(64) PFV_WAVES_PER_EU(kTfWaves) void k_pf_transform(FrameGeom g, const uint8_t *__restrict__ src, const uint8_t *__restrict__ ref,
                                                     const int8_t *__restrict__ mv, const uint8_t *__restrict__ has, int16_t *__restrict__ coef,
                                                     uint8_t *__restrict__ recon, const QTab *__restrict__ qtabs, float qmagic)
{
    __shared__ __attribute__((aligned(16))) int xw[kXchgDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kQTabDwords];
    __shared__ __attribute__((aligned(16))) int strip_tab[kTfStrips][4];   // per strip of the group: x0, y0, first macroblock (frame-relative)
    __shared__ int slot_tab[64];                                            // per coded macroblock, in group order: place (0..63) | cx << 8 | cy << 16
    __shared__ int pass_mbi[kStripMB];                                      // the pass's 8 macroblocks (frame-relative index, -1: empty slot)

    const int lane = threadIdx.x & 63, m = lane >> 3, i = lane & 7;
    // group -> stream, plane, first strip (all wave-uniform)
    const int gpf = tf_groups_per_frame(g);
    const int gq = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int stream = gq / gpf;
    int gr = gq - stream * gpf, plane = 0;
    for (; plane + 1 < g.n_planes && gr >= tf_groups_of_plane(g.p[plane]); plane++) gr -= tf_groups_of_plane(g.p[plane]);
    const PlaneGeom &p = g.p[plane];
    const int strips_p = p.strips_x * p.bh;
    const int s_first = gr * kTfStrips;

    // lane j <-> macroblock (j & 7) of strip s_first + (j >> 3)
    bool coded = false;
    int cxy = 0;
    {
        const int s = s_first + m;
        const int by = s / p.strips_x, sx = s - by * p.strips_x;
        const int n_mb = min(kStripMB, p.bw - sx * kStripMB), mb_first = p.mb0 + by * p.bw + sx * kStripMB;
        if (i == 0) reinterpret_cast<int4 *>(strip_tab[m])[0] = make_int4(sx * (kStripMB * 16), by * 16, mb_first, 0);
        if (s < strips_p && i < n_mb) {
            const long mbi = (long)stream * g.mbs_per_frame + mb_first + i;
            if (has[mbi]) {
                coded = true;
                cxy = ((int)mv[mbi * 2 + 0] & 0xff) << 8 | ((int)mv[mbi * 2 + 1] & 0xff) << 16;
            }
        }
    }
    const unsigned long long bal = __ballot(coded);
    const int n_coded = __builtin_popcountll(bal);
    if (n_coded == 0) return;
    if (coded) slot_tab[__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = lane | cxy;
    fill_qtable<true, FLT>(qtab_lds, qtabs + p.qsel, lane);
    wave_lds_sync();

    const LaneQ lq{qtab_lds, i};
    const uint8_t *splane = frame_src(g, src, stream) + p.src_off;
    const uint8_t *rplane = ref + (long)stream * g.pad_frame_bytes + p.pad_off;
    uint8_t *oplane = recon ? recon + (long)stream * g.pad_frame_bytes + p.pad_off : nullptr;
    int16_t *coef_frame = coef + (long)stream * g.mbs_per_frame * 256;
    for (int k0 = 0; k0 < n_coded; k0 += kStripMB) {
        const bool has_slot = k0 + m < n_coded;
        if (i == 0) {
            int mbf = -1;
            if (has_slot) {
                const int e = slot_tab[k0 + m];
                mbf = strip_tab[(e >> 3) & 7][2] + (e & 7);
            }
            pass_mbi[m] = mbf;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // the rows are fetched half by half: 8 registers of pixels alive instead of 16
            uint4 srow = make_uint4(0, 0, 0, 0), prow = srow;
            if (has_slot) {
                const int e = slot_tab[k0 + m];
                const int4 st = reinterpret_cast<const int4 *>(strip_tab[(e >> 3) & 7])[0];
                const int mbx = st.x + (e & 7) * 16, y = st.y + i + 8 * h;
                const int cx = (int)(int8_t)(e >> 8), cy = (int)(int8_t)(e >> 16);
                srow = load_src16(splane, p, mbx, y);
                prow = load_unaligned16(rplane + (long)(y + cy) * p.pw + (mbx + cx));
            }
            const uint4 o = penc_half<FLT, true>(srow, prow, xw, m, i, lq, qmagic, recon != nullptr, [&]() {
#pragma unroll
                for (int j = 0; j < 2; j++) {   // the stage's 128 16-byte chunks, 16 per slot, to the slots' macroblocks
                    const int ch = j * 64 + lane, mbf = pass_mbi[ch >> 4];
                    if (mbf >= 0)
                        st_stream(&reinterpret_cast<uint4 *>(coef_frame + (long)mbf * 256)[h * 16 + (ch & 15)], reinterpret_cast<const uint4 *>(xw)[stage_chunk(ch)]);
                }
            });
            if (oplane && has_slot) {
                // the macroblock's place is looked up again after the pipeline: nothing of it stays alive across the transforms
                const int e = slot_tab[k0 + m];
                const int4 st = reinterpret_cast<const int4 *>(strip_tab[(e >> 3) & 7])[0];
                *reinterpret_cast<uint4 *>(oplane + (long)(st.y + i + 8 * h) * p.pw + st.x + (e & 7) * 16) = o;
            }
        }
        wave_lds_sync();   // pass_mbi is rewritten by the next pass
    }
}

// ================================================================== P-frame encode, small-grid lane mapping (16 lanes per macroblock)
// The same tile decomposition as k_enc_pframe (128 x 64 pixels, one reference window in LDS), EIGHT wavefronts per workgroup:
// wavefront w owns half h = w & 1 of strip w >> 1, i.e. 4 macroblocks; lane (mb, r) = (lane >> 4, lane & 15) owns source row r of
// its macroblock for the search (half the dot products of the row-pair form per lane, the partial sums of a candidate come from 16
// lanes) and, as slot = lane >> 3 = (mb, r >> 3), one half of it for the transform -- one pass of the half-macroblock pipeline
// instead of two.  A wavefront's serial length roughly halves; that is what a launch lasts when it has only one or two
// wavefronts per SIMD (see "Lane mappings").  The exchange regions of wavefronts 0..3 live in the released window, those of
// 4..7 in the released reduction regions.  No tile compaction here (it serves throughput on full devices).
constexpr int kThreads16 = 64 * 2 * kStripsPerWG;       // 8 wavefronts
constexpr int kWaves16 = 2 * kStripsPerWG;
constexpr int kRedPitch16 = 144;                        // dwords per macroblock: 8 candidates x 16 lanes; 144 = 16 (mod 64): the 4 macroblocks of a wavefront
                                                        // write 64 different banks
static_assert(4 * kRedPitch16 <= kRedDwords, "the 16-lane reduction layout fits the reduction region");
static_assert(kStripsPerWG * kXchgDwords * 4 <= kWinAlloc && kStripsPerWG * kXchgDwords <= kWaves16 * kRedDwords, "exchange regions of 8 wavefronts");
__device__ __forceinline__ constexpr int win_first_issue16(int w) { return (w * kWinIssues + kWaves16 - 1) / kWaves16; }
constexpr int kRowRor8 = 0x128;                         // DPP row_ror:8: lane r <-> r ^ 8 inside each 16 lanes

__device__ __forceinline__ int mb_sum16(int v)
{
    v = mb_sum(v);
    return v + dpp<kRowRor8>(v);
}

struct SearchLane16 {
    int neg2, ord, dy, dx;
    int *wr;            // &red[mb][0][r]: the lane's partial for candidate c goes to wr[c * 16]
    const int4 *rd;     // &red[mb][c][8 * (r >> 3)]: eight of the sixteen partials of this lane's candidate c = r & 7
};
__device__ __forceinline__ SearchLane16 make_search_lane16(int *red, int mb, int r, int neg2)
{
    SearchLane16 sl;
    const int c = r & 7;
    sl.neg2 = neg2;
    sl.ord = c + 1;
    const int b9 = c < 4 ? c : c + 1;
    sl.dy = b9 / 3 - 1;
    sl.dx = b9 - (b9 / 3) * 3 - 1;
    sl.wr = red + mb * kRedPitch16 + r;
    sl.rd = reinterpret_cast<const int4 *>(red + mb * kRedPitch16 + c * 16 + (r & 8));
    return sl;
}

// One search level, one source row per lane (see search_level for the arithmetic: SSD = sum a^2 - 2 sum ab + sum b^2 in u32,
// accept rule = lexicographic minimum of (error, visiting order), src/common.rs:154-204).  wrow0: window row of the lane's row.
template <int S, bool FIRST, bool BOUNDS>
__device__ __forceinline__ void search_level16(const uint8_t *win, int wrow0, int wcol0, const uint4 &a, int a2, int mbx, int mby, int pw, int ph,
                                               SearchState &st, const SearchLane16 &sl)
{
    const int sh = (st.cx - 1) & 3;                // step 1: phase of the leftmost candidate
    const int col = S >= 4 ? (wcol0 + st.cx - S) : (S == 2 ? wcol0 + st.cx - 4 : wcol0 + ((st.cx - 1) & ~3));
    int part[3][3];
#pragma unroll
    for (int my = -1; my <= 1; my++) {
        const uint8_t *rp = win + (wrow0 + st.cy + my * S) * kWinStride + col;
        const bool centre_known = !FIRST && my == 0;   // (0,0) is not evaluated again (:176)
        unsigned ab[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
        if (S == 8) {          // col = 8 (mod 16): 8-, 16-, 8-byte pieces; candidates at dwords 0, 2, 4 of an 8-dword span
            const uint2 p0 = *reinterpret_cast<const uint2 *>(rp), p2 = *reinterpret_cast<const uint2 *>(rp + 24);
            const uint4 p1 = *reinterpret_cast<const uint4 *>(rp + 8);
            const unsigned d[8] = {p0.x, p0.y, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y};
            const unsigned q23 = sq2(d[2], d[3], 0), q45 = sq2(d[4], d[5], 0);
            bb[0] = sq2(d[0], d[1], q23); bb[1] = q23 + q45; bb[2] = sq2(d[6], d[7], q45);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                ab[c] = dot_ab(a, d[2 * c], d[2 * c + 1], d[2 * c + 2], d[2 * c + 3], 0);
            }
        } else if (S == 4) {   // dword, two 8-byte pieces, dword; candidates at dwords 0, 1, 2 of a 6-dword span
            const unsigned *t = reinterpret_cast<const unsigned *>(rp);
            const uint2 ta = *reinterpret_cast<const uint2 *>(rp + 4), tb = *reinterpret_cast<const uint2 *>(rp + 12);
            const unsigned d[6] = {t[0], ta.x, ta.y, tb.x, tb.y, t[5]};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                ab[c] = dot_ab(a, d[c], d[c + 1], d[c + 2], d[c + 3], 0);
            }
            const unsigned q23 = sq2(d[2], d[3], 0);
            if (centre_known) {
                bb[0] = sq2(d[0], d[1], q23);
                bb[2] = sq2(d[4], d[5], q23);
            } else {
                const unsigned y = __builtin_amdgcn_udot4(d[1], d[1], q23, false), z = __builtin_amdgcn_udot4(d[4], d[4], 0, false);
                bb[0] = __builtin_amdgcn_udot4(d[0], d[0], y, false);
                bb[1] = y + z;
                bb[2] = __builtin_amdgcn_udot4(d[5], d[5], q23 + z, false);
            }
        } else {
            const unsigned *t = reinterpret_cast<const unsigned *>(rp);
            unsigned rt[6], dT[5];
#pragma unroll
            for (int k = 0; k < 6; k++) rt[k] = t[k];
            if (S == 1) {
#pragma unroll
                for (int k = 0; k < 5; k++) dT[k] = __builtin_amdgcn_alignbyte(rt[k + 1], rt[k], sh);   // bytes from (cx - 1) + 4k
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (centre_known && c == 1) continue;
                unsigned e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (S == 2) {   // cx is a multiple of 4 after steps 8 and 4: candidates 2 bytes before, on, 2 bytes after a dword boundary
                        const int o = k + (c >> 1);
                        e[k] = c == 1 ? rt[k + 1] : __builtin_amdgcn_alignbyte(rt[o + 1], rt[o], 2);
                    } else {
                        e[k] = c ? __builtin_amdgcn_alignbyte(dT[k + 1], dT[k], c) : dT[k];
                    }
                }
                ab[c] = dot_ab(a, e[0], e[1], e[2], e[3], 0);
                bb[c] = sq4(e[0], e[1], e[2], e[3], 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) part[my + 1][c] = __mul24((int)ab[c], sl.neg2) + (int)bb[c];
    }
    int ord = 0;
#pragma unroll
    for (int my = -1; my <= 1; my++) {
#pragma unroll
        for (int mx = -1; mx <= 1; mx++) {
            if (my == 0 && mx == 0) continue;
            sl.wr[ord * 16] = part[my + 1][mx + 1];
            ord++;
        }
    }
    wave_lds_sync();
    const int4 x = sl.rd[0], y = sl.rd[1];
    wave_lds_sync();
    const int half = ((x.x + x.y) + (x.z + x.w)) + ((y.x + y.y) + (y.z + y.w));
    const int err = a2 + half + dpp<kRowRor8>(half);
    bool valid = true;
    if (BOUNDS) {
        const int oy = mby + st.cy + sl.dy * S, ox = mbx + st.cx + sl.dx * S;
        valid = oy >= 0 && oy <= ph - 16 && ox >= 0 && ox <= pw - 16;        // :171, :182
    }
    const unsigned key = valid ? (((unsigned)err << 4) | (unsigned)sl.ord) : 0xffffffffu;
    const unsigned centre = FIRST ? ((unsigned)(a2 + mb_sum16(part[1][1])) << 4) : ((unsigned)st.err << 4);
    const unsigned best = min(mb_min(key), centre);
    const int bo = (int)(best & 15u);
    st.err = (int)(best >> 4);
    st.cy += ((int)((0x2A501u >> (2 * bo)) & 3u) - 1) * S;
    st.cx += ((int)((0x24891u >> (2 * bo)) & 3u) - 1) * S;
}

template <bool FLT>
__global__ __launch_bounds__(kThreads16) void k_enc_pframe16(FrameGeom g, const uint8_t *__restrict__ src, const uint8_t *__restrict__ ref,
                                                            int8_t *__restrict__ mv_out, uint8_t *__restrict__ has_out, int16_t *__restrict__ coef,
                                                            uint8_t *__restrict__ recon, const QTab *__restrict__ qtabs, float min_err, int neg2,
                                                            float qmagic)
{
    __shared__ __attribute__((aligned(16))) uint8_t win_lds[16 + kWinAlloc];
    __shared__ __attribute__((aligned(16))) int red_lds[kWaves16][kRedDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kQTabDwords];
    uint8_t *win = win_lds + 16;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strip = wave >> 1, half_strip = wave & 1;
    const int slot = lane >> 3, i = lane & 7, mb = lane >> 4, r = lane & 15;
    const int m = half_strip * 4 + mb;                      // macroblock within the strip
    const int vt = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const TilePos cur = locate_tile(g, vt, strip);
    const PlaneGeom &p = g.p[cur.sp.plane];
    if (wave == 0) fill_qtable<true, FLT>(qtab_lds, qtabs + p.qsel, lane);
    {   // the window: 17 wave-wide 1 KiB loads dealt to 8 wavefronts (see issue_window)
        const uint8_t *refp = ref + (long)cur.sp.stream * g.pad_frame_bytes + p.pad_off;
        for (int q = win_first_issue16(wave); q < win_first_issue16(wave + 1); q++) {
            const int c = q * 64 + lane;
            const int row = c / kWinChunksPerRow, col = c - row * kWinChunksPerRow;
            const int y = min(max(cur.winy0 + row, 0), p.ph - 1), x = min(max(cur.winx0 + col * 16, 0), p.pw - 16);
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(refp + (long)y * p.pw + x), (lds_void_t *)(win + c * 16), 16, 0, 0);
        }
    }
    const bool wave_valid = cur.wave_valid && half_strip * 4 < cur.sp.n_mb;
    const bool mb_valid = wave_valid && m < cur.sp.n_mb;
    uint4 row = make_uint4(0, 0, 0, 0);
    if (wave_valid) row = load_src16(frame_src(g, src, cur.sp.stream) + p.src_off, p, cur.sp.x0 + m * 16, cur.sp.y0 + r);
    __syncthreads();   // window complete
    int cx = 0, cy = 0;
    bool coded = false;
    uint4 patch = make_uint4(0, 0, 0, 0);
    if (wave_valid) {
        const int mbx = cur.sp.x0 + m * 16, mby = cur.sp.y0;
        const int wcol0 = mbx - cur.winx0, wrow0 = mby - cur.winy0 + r;
        unsigned s2 = __builtin_amdgcn_udot4(row.w, row.w, __builtin_amdgcn_udot4(row.z, row.z, __builtin_amdgcn_udot4(row.y, row.y, __builtin_amdgcn_udot4(row.x, row.x, 0, false), false), false), false);
        const int a2 = mb_sum16((int)s2);
        SearchState st;
        st.cx = 0; st.cy = 0; st.err = 0;
        const SearchLane16 sl = make_search_lane16(red_lds[wave], mb, r, neg2);
        // the whole half strip at least 15 px inside the plane: no bounds tests (wave-uniform)
        const int hx0 = cur.sp.x0 + half_strip * 64;
        const bool interior = hx0 >= 16 && hx0 + 64 + 16 <= p.pw && cur.sp.y0 >= 16 && cur.sp.y0 + 32 <= p.ph;
        if (interior) {
            search_level16<8, true, false>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<4, false, false>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<2, false, false>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<1, false, false>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
        } else {
            search_level16<8, true, true>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<4, false, true>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<2, false, true>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
            search_level16<1, false, true>(win, wrow0, wcol0, row, a2, mbx, mby, p.pw, p.ph, st, sl);
        }
        coded = mb_valid && !((float)st.err <= min_err);     // skip decision (src/common.rs:209, :221), compared in f32
        cx = st.cx; cy = st.cy;
        const int wx = wcol0 + cx, shp = wx & 3;           // the lane's row of the chosen patch (get_block of the reconstruction, :261)
        const unsigned *d = reinterpret_cast<const unsigned *>(win + (wrow0 + cy) * kWinStride + (wx & ~3));
        const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
        patch = make_uint4(__builtin_amdgcn_alignbyte(d1, d0, shp), __builtin_amdgcn_alignbyte(d2, d1, shp), __builtin_amdgcn_alignbyte(d3, d2, shp),
                           __builtin_amdgcn_alignbyte(d4, d3, shp));
    }
    const long mbi = (long)cur.sp.stream * g.mbs_per_frame + cur.sp.mb_first + m;
    if (r == 0 && mb_valid) {
        mv_out[mbi * 2 + 0] = (int8_t)cx;
        mv_out[mbi * 2 + 1] = (int8_t)cy;
        has_out[mbi] = coded ? 1 : 0;
    }
    __syncthreads();   // window and reduction regions released by every wavefront
    if (!wave_valid) return;
    int *xw = wave < kStripsPerWG ? reinterpret_cast<int *>(win) + wave * kXchgDwords : &red_lds[0][0] + (wave - kStripsPerWG) * kXchgDwords;
    int16_t *coef_mb0 = coef + ((long)cur.sp.stream * g.mbs_per_frame + cur.sp.mb_first) * 256;
    uint8_t *dst = recon ? recon + (long)cur.sp.stream * g.pad_frame_bytes + p.pad_off + (long)(cur.sp.y0 + r) * p.pw + cur.sp.x0 + m * 16 : nullptr;
    if (__any(coded)) {
        const LaneQ lq{qtab_lds, i};
        const uint4 o = penc_half<FLT>(coded ? row : patch, patch, xw, slot, i, lq, qmagic, recon != nullptr,
                                       [&]() { store_coef_quads(xw, coef_mb0, cur.sp.n_mb, lane, half_strip); });
        if (recon && mb_valid) *reinterpret_cast<uint4 *>(dst) = o;
    } else if (mb_valid) {   // the 16 lanes of a skipped macroblock: zero coefficients (2 x 16 bytes each), prediction as reconstruction
        uint4 *cm = reinterpret_cast<uint4 *>(coef_mb0 + m * 256);
        st_stream(&cm[r], make_uint4(0, 0, 0, 0));
        st_stream(&cm[16 + r], make_uint4(0, 0, 0, 0));
        if (dst) *reinterpret_cast<uint4 *>(dst) = patch;
    }
}

// ================================================================== I-frame decode
// reference: VideoPlane::decode_plane / decode_plane_into (src/common.rs:423-446, 477-496)
// frames_out != nullptr: also write the cropped, tightly packed retframe (n_streams frames).
template <int LPM = 8, bool LISTS = false>
__global__ __launch_bounds__(kThreads) void k_dec_iframe(FrameGeom g, const int16_t *__restrict__ coef, CoefLists cl,
                                                          uint8_t *__restrict__ out, const QTab *__restrict__ qtabs,
                                                          uint8_t *__restrict__ frames_out)
{
    __shared__ __attribute__((aligned(16))) int xchg[kStripsPerWG][kXchgDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kStripsPerWG][kQTabDwords];
    constexpr int kPasses = LPM == 8 ? 2 : 1;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPRs
    const int gw = xcd_remap((int)blockIdx.x, (int)gridDim.x) * kStripsPerWG + wave;
    const int gstrip = LPM == 8 ? gw : gw >> 1, half_strip = LPM == 8 ? 0 : gw & 1;     // see "Lane mappings"
    if (gstrip >= g.strips_per_frame * g.n_streams) return;
    const StripPos sp = locate_strip(g, gstrip);
    if (half_strip * 4 >= sp.n_mb) return;
    const PlaneGeom &p = g.p[sp.plane];
    const int slot = lane >> 3, i = lane & 7;
    const int m = LPM == 8 ? slot : half_strip * 4 + (slot >> 1);
    int *xw = xchg[wave];

    const int16_t *coef_mb0 = coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256;
    uint4 cbuf[kPasses][2];
    ListSpan span;
    if (LISTS) {
        const int last = min(sp.n_mb - half_strip * 4, LPM == 8 ? 8 : 4) - 1;     // the wavefront's last macroblock that exists (0-based, within the wavefront)
        const uint32_t *cnt = cl.counts + (long)sp.stream * (g.mbs_per_frame + 1) + sp.mb_first + half_strip * 4;
        const int mw = min(LPM == 8 ? slot : (slot >> 1), last);
        span = list_span(cl.entries[sp.stream], cnt[mw], cnt[mw + 1], last * (LPM == 8 ? 8 : 16), lane);
    } else if (LPM == 8) {
        fetch_coef_half(cbuf[0], coef_mb0, sp.n_mb, lane, 0);
        fetch_coef_half(cbuf[kPasses - 1], coef_mb0, sp.n_mb, lane, 1);
    } else {
        fetch_coef_quads(cbuf[0], coef_mb0, sp.n_mb, lane, half_strip);
    }
    fill_qtable<false>(qtab_lds[wave], qtabs + p.qsel, lane);
    wave_lds_sync();
    const LaneQ lq{qtab_lds[wave], i};
    uint8_t *dst = out + (long)sp.stream * g.pad_frame_bytes + p.pad_off + (long)(sp.y0 + i) * p.pw + sp.x0 + m * 16;
#pragma unroll
    for (int pass = 0; pass < kPasses; pass++) {
        const int h = LPM == 8 ? pass : (slot & 1);
        if (LISTS) stage_list_half<LPM>(xw, span, lane, sp.mb_first + half_strip * 4, pass);
        else stage_coef_half(xw, cbuf[pass], lane);
        wave_lds_sync();
        int v[2][8];
        gather_half(v, xw, slot, lq);
        inverse_half(v, xw, slot, i, lq);
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int k = 0; k < 8; k++) v[s][k] = min(max(v[s][k] + 128, 0), 255);   // src/common.rs:321
        if (m < sp.n_mb) {
            const uint4 o = pack_row(v);
            *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = o;
            if (frames_out)
                store_cropped16(frames_out + (long)sp.stream * g.src_frame_bytes + p.src_off, p, sp.x0 + m * 16, sp.y0 + i + 8 * h, o);
        }
    }
}

// ================================================================== P-frame decode
// reference: VideoPlane::decode_plane_delta / _into (src/common.rs:448-475, 498-521) ->
// decode_block_delta (:254-285).  ref and out are distinct buffers (ping-pong): the
// reference reads every patch from the old plane before it writes anything (:498-521).
// err_flag[stream] is set when a motion vector leaves the plane (:258-259 debug_assert); the
// vector is then treated as (0,0) so that no out-of-bounds access happens.
template <int LPM = 8, bool LISTS = false>
__global__ __launch_bounds__(kThreads) void k_dec_pframe(FrameGeom g, const int8_t *__restrict__ mv,
                                                          const uint8_t *__restrict__ has, const int16_t *__restrict__ coef, CoefLists cl,
                                                          const uint8_t *__restrict__ ref, uint8_t *__restrict__ out,
                                                          const QTab *__restrict__ qtabs, int *__restrict__ err_flag,
                                                          uint8_t *__restrict__ frames_out)
{
    __shared__ __attribute__((aligned(16))) int xchg[kStripsPerWG][kXchgDwords];
    __shared__ __attribute__((aligned(16))) int qtab_lds[kStripsPerWG][kQTabDwords];
    constexpr int kPasses = LPM == 8 ? 2 : 1;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform -> SGPRs
    const int gw = xcd_remap((int)blockIdx.x, (int)gridDim.x) * kStripsPerWG + wave;
    const int gstrip = LPM == 8 ? gw : gw >> 1, half_strip = LPM == 8 ? 0 : gw & 1;     // see "Lane mappings"
    if (gstrip >= g.strips_per_frame * g.n_streams) return;
    const StripPos sp = locate_strip(g, gstrip);
    if (half_strip * 4 >= sp.n_mb) return;
    const PlaneGeom &p = g.p[sp.plane];
    const int slot = lane >> 3, i = lane & 7;
    const int m = LPM == 8 ? slot : half_strip * 4 + (slot >> 1);
    int *xw = xchg[wave];
    const bool mb_valid = m < sp.n_mb;
    const long mbi = (long)sp.stream * g.mbs_per_frame + sp.mb_first + (mb_valid ? m : 0);

    // first round trip: block headers and (independent of them) the quantiser constants
    int mx = mv[mbi * 2 + 0], my = mv[mbi * 2 + 1];
    const bool coded = mb_valid && has[mbi] != 0;
    uint32_t cnt0 = 0, cnt1 = 0;
    int last_mb = 0;
    if (LISTS) {   // the macroblock's share of the stream's list (no dependent round trip: read beside the block headers)
        last_mb = min(sp.n_mb - half_strip * 4, LPM == 8 ? 8 : 4) - 1;
        const uint32_t *cnt = cl.counts + (long)sp.stream * (g.mbs_per_frame + 1) + sp.mb_first + half_strip * 4;
        const int mw = min(LPM == 8 ? slot : (slot >> 1), last_mb);
        cnt0 = cnt[mw]; cnt1 = cnt[mw + 1];
    }
    fill_qtable<false>(qtab_lds[wave], qtabs + p.qsel, lane);

    const int mbx = sp.x0 + m * 16, mby = sp.y0;
    if (mb_valid) {
        int sx = mbx + mx, sy = mby + my;
        if (sx < 0 || sx > p.pw - 16 || sy < 0 || sy > p.ph - 16) {
            if (i == 0) atomicOr(err_flag + sp.stream, 1);   // one flag per stream of the launch
            mx = 0; my = 0;
        }
    }
    // second round trip: coefficients (only if some macroblock of the wavefront has any) and patch rows
    const bool any_coded = __any(coded);
    const int16_t *coef_mb0 = coef + ((long)sp.stream * g.mbs_per_frame + sp.mb_first) * 256;
    uint4 cbuf[kPasses][2];
    ListSpan span;
    if (any_coded) {
        if (LISTS) {
            span = list_span(cl.entries[sp.stream], cnt0, cnt1, last_mb * (LPM == 8 ? 8 : 16), lane);
        } else if (LPM == 8) {
            fetch_coef_half(cbuf[0], coef_mb0, sp.n_mb, lane, 0);
            fetch_coef_half(cbuf[kPasses - 1], coef_mb0, sp.n_mb, lane, 1);
        } else {
            fetch_coef_quads(cbuf[0], coef_mb0, sp.n_mb, lane, half_strip);
        }
    }
    uint4 patch[kPasses];
#pragma unroll
    for (int pass = 0; pass < kPasses; pass++) patch[pass] = make_uint4(0, 0, 0, 0);
    if (mb_valid) {   // the lane's rows of the motion-compensated patch (get_block, :327-339)
        const uint8_t *rp = ref + (long)sp.stream * g.pad_frame_bytes + p.pad_off + (long)(mby + my + i) * p.pw + (mbx + mx);
#pragma unroll
        for (int pass = 0; pass < kPasses; pass++) patch[pass] = load_unaligned16(rp + 8 * (long)(LPM == 8 ? pass : (slot & 1)) * p.pw);
    }
    uint8_t *dst = out + (long)sp.stream * g.pad_frame_bytes + p.pad_off + (long)(sp.y0 + i) * p.pw + mbx;
    uint8_t *crop = frames_out ? frames_out + (long)sp.stream * g.src_frame_bytes + p.src_off : nullptr;

    if (any_coded) {
        wave_lds_sync();
        const LaneQ lq{qtab_lds[wave], i};
#pragma unroll
        for (int pass = 0; pass < kPasses; pass++) {
            const int h = LPM == 8 ? pass : (slot & 1);
            if (LISTS) stage_list_half<LPM>(xw, span, lane, sp.mb_first + half_strip * 4, pass);
            else stage_coef_half(xw, cbuf[pass], lane);
            wave_lds_sync();
            int v[2][8], pp[2][8];
            const int codedmask = coded ? -1 : 0;
            gather_half(v, xw, slot, lq);
            inverse_half(v, xw, slot, i, lq);
            unpack_row(patch[pass], pp);
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int k = 0; k < 8; k++)   // apply_residuals (:98-104); masked to 0 for skipped blocks: copy
                    pp[s][k] = min(max(pp[s][k] + 2 * (v[s][k] & codedmask), 0), 255);
            if (mb_valid) {
                const uint4 o = pack_row(pp);
                *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = o;
                if (crop) store_cropped16(crop, p, mbx, sp.y0 + i + 8 * h, o);
            }
        }
    } else if (mb_valid) {
#pragma unroll
        for (int pass = 0; pass < kPasses; pass++) {
            const int h = LPM == 8 ? pass : (slot & 1);
            *reinterpret_cast<uint4 *>(dst + (long)(8 * h) * p.pw) = patch[pass];
            if (crop) store_cropped16(crop, p, mbx, sp.y0 + i + 8 * h, patch[pass]);
        }
    }
}

#ifndef PFV_PENC_TU   // the non-template kernels exist once: in the main translation unit
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

// VideoPlane::reduce (src/common.rs:523-536): point-sampled 2x decimation (every second pixel of every second row),
// used by VideoFrame::from_planes (src/frame.rs:51-59) to make 4:2:0 chroma.  dst is (src_w/2) x (src_h/2).
__global__ __launch_bounds__(kThreads) void k_reduce2x(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int src_w,
                                                        int src_h)
{
    const int dw = src_w >> 1, dh = src_h >> 1;
    const long n = (long)dw * dh;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        int y = (int)(idx / dw), x = (int)(idx - (long)y * dw);
        dst[idx] = src[(long)(2 * y) * src_w + 2 * x];
    }
}
// VideoPlane::double (src/common.rs:538-556): nearest-neighbour 2x upsampling.  dst is (2 src_w) x (2 src_h).
__global__ __launch_bounds__(kThreads) void k_double2x(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int src_w,
                                                        int src_h)
{
    const int dw = src_w * 2;
    const long n = (long)dw * src_h * 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        int y = (int)(idx / dw), x = (int)(idx - (long)y * dw);
        dst[idx] = src[(long)(y >> 1) * src_w + (x >> 1)];
    }
}

// f32 -> u8 the way Rust's `as u8` does it: truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ uint32_t f32_as_u8(float x)
{
    return x >= 255.0f ? 255u : (x > 0.0f ? (uint32_t)(int)x : 0u);
}
// load_frame of the reference's tests (src/lib.rs:337-359) + VideoFrame::from_planes (src/frame.rs:51-59): interleaved
// RGB8 -> Y at full resolution, U and V from the pixels at even (x, y) (from_planes point-samples with reduce()).
// JPEG-conversion YCbCr in f32, evaluated left to right without contraction, then `as u8`.
__global__ __launch_bounds__(kThreads) void k_rgb_to_yuv420(const uint8_t *__restrict__ rgb, int w, int h, uint8_t *__restrict__ frame)
{
#pragma clang fp contract(off)
    const long n = (long)w * h;
    const int cw = w >> 1, ch = h >> 1;
    uint8_t *py = frame, *pu = frame + n, *pv = pu + (long)cw * ch;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / w), x = (int)(idx - (long)y * w);
        const float r = (float)rgb[3 * idx], g = (float)rgb[3 * idx + 1], b = (float)rgb[3 * idx + 2];
        py[idx] = (uint8_t)f32_as_u8((0.299f * r) + (0.587f * g) + (0.114f * b));
        if (!(x & 1) && !(y & 1) && (x >> 1) < cw && (y >> 1) < ch) {
            const long c = (long)(y >> 1) * cw + (x >> 1);
            pu[c] = (uint8_t)f32_as_u8(128.0f - (0.168736f * r) - (0.331264f * g) + (0.5f * b));
            pv[c] = (uint8_t)f32_as_u8(128.0f + (0.5f * r) - (0.418688f * g) - (0.081312f * b));
        }
    }
}
// save_frame (src/lib.rs:361-394): chroma doubled (nearest), JPEG-conversion YCbCr -> RGB8 in f32, `as u8`.
__global__ __launch_bounds__(kThreads) void k_yuv420_to_rgb(const uint8_t *__restrict__ frame, int w, int h, uint8_t *__restrict__ rgb)
{
#pragma clang fp contract(off)
    const long n = (long)w * h;
    const int cw = w >> 1, ch = h >> 1;
    const uint8_t *py = frame, *pu = frame + n, *pv = pu + (long)cw * ch;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / w), x = (int)(idx - (long)y * w);
        const long c = (long)(y >> 1) * cw + (x >> 1);
        const float yy = (float)py[idx], u = (float)pu[c] - 128.0f, v = (float)pv[c] - 128.0f;
        rgb[3 * idx] = (uint8_t)f32_as_u8(yy + (1.402f * v));
        rgb[3 * idx + 1] = (uint8_t)f32_as_u8(yy - (0.344136f * u) - (0.714136f * v));
        rgb[3 * idx + 2] = (uint8_t)f32_as_u8(yy + (1.772f * u));
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

// Sparse coefficient upload (pfv_dec_*_sparse): coef[idx[i]] = val[i] onto a zeroed coefficient buffer.  Indices of one
// frame are distinct and ascending (the bit parser walks the frame front to back), so neighbouring lanes hit
// neighbouring lines.
__global__ void __launch_bounds__(kThreads) k_scatter_coef(const uint32_t *idx, const int16_t *val, uint32_t n, uint32_t limit, int16_t *coef)
{
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const uint32_t at = idx[i];
        if (at < limit) coef[at] = val[i];
    }
}

// The same for n_streams lists at once (pfv_batch_decoder): stream k's pairs are idx[k * cap .. k * cap + counts[k]); indices are
// already absolute.  The lists may live in page-locked HOST memory: the kernel then reads them over PCIe (coalesced, read once)
// and no staging copy is needed.
__global__ void __launch_bounds__(kThreads) k_scatter_coef_seg(const uint32_t *idx, const int16_t *val, const uint32_t *counts, uint32_t cap,
                                                               uint32_t limit, int16_t *coef)
{
    const uint32_t k = blockIdx.y, n = counts[k];
    const uint32_t *ik = idx + (size_t)k * cap;
    const int16_t *vk = val + (size_t)k * cap;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const uint32_t at = ik[i];
        if (at < limit) coef[at] = vk[i];
    }
}

#endif  // PFV_PENC_TU

}  // namespace pfv
