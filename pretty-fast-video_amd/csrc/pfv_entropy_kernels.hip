// pfv_entropy_kernels.hip -- the encoder's entropy stage on the device (gfx950).
//
// The reference serialises a frame on one host thread (src/enc.rs:237-320 i-frames, :332-470 p-frames): rle_encode per
// macroblock (src/rle.rs:9-47), one 16-symbol histogram per frame (rle.rs:40-47), a Huffman tree from the normalised
// histogram (rle.rs:49-66, src/huffman.rs:71-119) and LSB-first bit packing.  At device rates that host stage is the
// whole cost of the encoder (a 1080p frame: ~30 us of kernels vs milliseconds of host entropy + 6 MB of PCIe), so
// the same byte stream is produced here, from the coefficient / header buffers the encode kernels left in HBM:
//
//   k_ent_scan   workgroup = 256 consecutive subblocks of one stream (64 macroblocks), lane = one 8x8 subblock.  The
//                group's 32 KiB of coefficients come in with coalesced 16-byte loads -- p-frames: only the macroblocks
//                that have any -- and are staged in LDS (rows padded to 136 B); each lane builds its non-zero bitmap
//                (v_pk_min_u16 + v_dot2_u32_u16 per coefficient pair) and walks the set bits: one 32-bit word per run
//                symbol (fillers, num_zeroes, coeff_size, value bits) appended to the group's symbol list -- the
//                lanes' runs are contiguous in it --, symbol counts in 16 byte-wide LDS counters, sum of coefficient
//                sizes.  Per workgroup -> HBM (symbol list and its length, counts, size sum, block-header bits); per
//                stream -> the frame histogram (atomics).
//   k_ent_codes  one workgroup per stream.  Wavefront 0 builds the reference's Huffman tree lane-parallel (stable
//                rank sort, ballot-positioned merges, codes by walking the parent chain), all lanes fill the 256
//                pre-joined (num_zeroes, coeff_size) code pairs; then bits per workgroup-of-scan = counts . code lengths
//                + sizes, exclusive prefix over the workgroups -> base bit offsets and the payload size.
//   k_ent_init   zeroes exactly the words the payload will occupy and writes the 19 header bytes.
//   k_ent_pack   one workgroup per group of k_ent_scan, lane = one SYMBOL of the group's list, 256 at a time (never the
//                coefficients): code pair and bit length from LDS tables, workgroup prefix sum of the lengths on top
//                of the group's base = the symbol's bit offset, bits ORed into an LDS window that leaves with
//                coalesced stores.
//
// Run order inside a macroblock is the coefficient buffer's own (zigzag within a subblock, subblocks 0..3), runs cross
// subblock boundaries and end at the macroblock (enc.rs:246-255), so lane (mb, sb) needs only the bitmaps of the
// earlier subblocks of its macroblock (its quad).  Included by pfv_capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfv_prof.h"   // ENT_MARK: phase timestamps of the experiment builds, empty otherwise

namespace pfv {

constexpr int kEntThreads = 256;                    // subblocks per workgroup of k_ent_scan / k_ent_pack
constexpr int kEntRow64 = 17;                       // LDS row of one subblock: 128 B of coefficients + 8 B pad, in 8-byte units
constexpr int kEntGroupSyms = kEntThreads * 65;      // symbol words one group can produce: 64 values + the closing run, per lane
constexpr uint32_t kEntErrOversize = 0xffffffffu;   // a coefficient needs more than 15 size bits (rle.rs:44 would panic)
constexpr uint32_t kEntErrCapacity = 0xfffffffeu;   // payload larger than the per-stream capacity

struct EntCodes {
    uint32_t pair_bits[256];   // [num_zeroes | coeff_size << 4]: code(num_zeroes) then code(coeff_size), LSB first
    uint8_t pair_len[256];
    uint8_t len[16];
    uint8_t table[16];         // the packet's 16 table bytes (rle.rs:49-66)
    uint32_t oversize;         // set by k_ent_scan
    uint32_t pad[3];
};

// what one workgroup of k_ent_scan found in its 256 subblocks
struct EntGroup {
    uint32_t counts[8];        // symbol counts, two 16-bit fields per word (symbol 2j low, 2j+1 high)
    uint32_t sumsize;          // sum of coeff_size over all values
    uint32_t hdr_bits;         // block-header bits of its macroblocks (p-frames)
    uint32_t sym_base;         // written by k_ent_codes: bit offset of the group's first symbol ...
    uint32_t hdr_base;         // ... and of its first block header
    uint32_t n_syms;           // words in the group's symbol list
};

struct EntFrame {
    int total_blocks;          // macroblocks per stream (Y then U then V)
    int n_streams;
    int n_groups;              // workgroups of k_ent_scan per stream = ceil(total_blocks * 4 / 256)
    int pframe;                // block headers present, has_coef honoured
    uint32_t cap_bytes;        // payload capacity per stream (multiple of 4)
    uint8_t qidx[3];
    uint8_t pad;
    uint32_t ones16;           // 0x00010001, from the host: a literal would let the compiler rewrite min(x, 1) per 16-bit half
                               // into compare + select + repack (6 instructions where v_pk_min_u16 is 1)
};

struct EntBufs {
    const int16_t *coef;       // [S][total_blocks][256]
    const int8_t *mv;          // [S][total_blocks][2]   (p-frames)
    const uint8_t *has;        // [S][total_blocks]      (p-frames)
    uint32_t *syms;            // [S][n_groups][kEntGroupSyms] run symbols: fillers << 24 | value bits << 8 | coeff_size << 4 | num_zeroes
    EntGroup *groups;          // [S][n_groups]
    int32_t *hist;             // [S][16]
    EntCodes *codes;           // [S]
    uint32_t *sizes;           // [S] payload bytes or kEntErr*
    uint8_t *payload;          // [S][cap_bytes]
};

__device__ __forceinline__ void ent_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t ent_shfl(uint32_t v, int src_lane)
{
    return (uint32_t)__shfl((int)v, src_lane);
}
__device__ __forceinline__ uint32_t ent_wave_sum(uint32_t v)
{
    for (int m = 1; m < 64; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
// inclusive prefix sum over the wavefront, DPP form (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31):
// six v_add with a DPP operand instead of six ds_bpermute + compare + add
__device__ __forceinline__ uint32_t ent_wave_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Position of the last non-zero coefficient before subblock `sb` in macroblock order (-1: none), from the quad's bitmaps.
template <int J>
__device__ __forceinline__ uint64_t ent_quad_bcast(uint64_t v)   // lane J of the quad, DPP quad_perm [J, J, J, J]
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, J * 0x55, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), J * 0x55, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int ent_prev_last(uint64_t mine, int sb)
{
    const uint64_t m0 = ent_quad_bcast<0>(mine), m1 = ent_quad_bcast<1>(mine), m2 = ent_quad_bcast<2>(mine);
    int last = -1;
    if (0 < sb && m0) last = 63 - __builtin_clzll(m0);
    if (1 < sb && m1) last = 64 + 63 - __builtin_clzll(m1);
    if (2 < sb && m2) last = 128 + 63 - __builtin_clzll(m2);
    return last;
}

// fillers (15, 0) needed before a run of `run` zeros can be coded in 4 bits, and what is left (rle.rs:18-21, 31-34)
__device__ __forceinline__ void ent_split_run(unsigned run, unsigned &fillers, unsigned &rest)
{
    fillers = run > 15u ? (run - 1u) / 15u : 0u;
    rest = run - 15u * fillers;
}

// The workgroup's 256 subblocks (32 KiB, contiguous in the coefficient buffer) -> LDS rows, coalesced: a half-wavefront
// moves one macroblock (512 B = 32 pieces of 16 bytes) per step, thread t the pieces of macroblocks 8 * (t / 32) + k,
// k = 0..7.  `has` (p-frames; nullptr for i-frames) points at the group's first macroblock flag: a macroblock without
// coefficients is never looked at by its lanes (`coded` in k_ent_scan), so its 512 bytes stay in HBM -- a third of a
// typical p-frame -- and its LDS rows stay unwritten.
__device__ __forceinline__ void ent_stage_rows(uint64_t *rows, const int16_t *group_base, int n_live_sb, const uint8_t *has)
{
    const uint4 *src = (const uint4 *)group_base;
    const int half = (int)(threadIdx.x >> 5), j = (int)(threadIdx.x & 31u);
    const int n_live_mb = n_live_sb >> 2;
    uint64_t flags = ~0ull;   // byte k: macroblock 8 * half + k has coefficients
    if (has) {
        if (n_live_sb == kEntThreads) {
            __builtin_memcpy(&flags, has + 8 * half, 8);
        } else {
            flags = 0;
            for (int k = 0; k < 8; k++)
                if (8 * half + k < n_live_mb) flags |= (uint64_t)has[8 * half + k] << (8 * k);
        }
    }
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++)   // all loads in flight before the first LDS write
        if (8 * half + k < n_live_mb && ((flags >> (8 * k)) & 0xffu)) v[k] = src[32 * (8 * half + k) + j];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (8 * half + k < n_live_mb && ((flags >> (8 * k)) & 0xffu)) {
            const int piece = 32 * (8 * half + k) + j, sbl = piece >> 3, part = piece & 7;
            rows[sbl * kEntRow64 + 2 * part] = (uint64_t)v[k].x | ((uint64_t)v[k].y << 32);
            rows[sbl * kEntRow64 + 2 * part + 1] = (uint64_t)v[k].z | ((uint64_t)v[k].w << 32);
        }
    }
}

// v_pk_min_u16: unsigned minimum of the two 16-bit halves (generic vector form; hipcc selects v_pk_min_u16 for it)
typedef unsigned short ent_us2 __attribute__((vector_size(4)));
__device__ __forceinline__ uint32_t ent_pk_min_u16(uint32_t a, uint32_t b)
{
    ent_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const ent_us2 m = x < y ? x : y;
    uint32_t r;
    __builtin_memcpy(&r, &m, 4);
    return r;
}
// v_dot2_u32_u16: acc + a.lo * b.lo + a.hi * b.hi
__device__ __forceinline__ uint32_t ent_udot2(uint32_t a, uint32_t b, uint32_t acc)
{
    ent_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_udot2(x, y, acc, false);
}
// Non-zero bitmap of 16 coefficients (8 dwords, two 16-bit values each) in the low 16 bits: v_pk_min_u16 against
// ones16 = (1, 1) turns each half into 0 / 1, v_dot2_u32_u16 drops the pair onto bits 2j, 2j + 1 of the accumulator
// -- two instructions per dword.
__device__ __forceinline__ uint32_t ent_nz16(const uint32_t (&d)[8], uint32_t ones16)
{
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) acc = ent_udot2(ent_pk_min_u16(d[j], ones16), (1u << (2 * j)) | (1u << (2 * j + 17)), acc);
    return acc;
}

// sum over each 16-lane row, valid in every lane of the row (DPP butterflies)
__device__ __forceinline__ uint32_t ent_row_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);   // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------------- k_ent_scan
__global__ void __launch_bounds__(kEntThreads) k_ent_scan(EntFrame f, EntBufs b)
{
    __shared__ uint64_t rows[kEntThreads * kEntRow64];
    __shared__ uint4 cnt4[kEntThreads];
    __shared__ uint32_t part[kEntThreads / 16][10];   // per 16-lane row: 8 count words, size sum, header bits
    __shared__ uint32_t wave_syms[kEntThreads / 64];
    const int stream = (int)blockIdx.y, n_sb = f.total_blocks * 4;
    const int sb0 = (int)blockIdx.x * kEntThreads;
    const int sbi = sb0 + (int)threadIdx.x;
    const bool live = sbi < n_sb;
    const int mb = sbi >> 2, sb = sbi & 3;
    ENT_MARK0();
    ENT_MARK(0, 0);
    const size_t bi = (size_t)stream * f.total_blocks + (live ? mb : 0);
    const bool coded = live && (!f.pframe || b.has[bi] != 0);   // in flight together with the coefficients, and so are the
    int mvx = 0, mvy = 0;                                       // motion vector components for the block-header bits
    if (f.pframe && live && sb == 0) { mvx = b.mv[2 * bi]; mvy = b.mv[2 * bi + 1]; }
    ent_stage_rows(rows, b.coef + ((size_t)stream * f.total_blocks * 4 + sb0) * 64, min(kEntThreads, n_sb - sb0),
                   f.pframe ? b.has + (size_t)stream * f.total_blocks + (sb0 >> 2) : nullptr);
    ENT_MARK(0, 1);
    __syncthreads();
    ENT_MARK(0, 2);

    const uint64_t *row = rows + threadIdx.x * kEntRow64;
    uint64_t mask = 0;
    if (coded) {
        uint32_t q[4];
#pragma unroll
        for (int fld = 0; fld < 4; fld++) {   // coefficients 16 * fld .. 16 * fld + 15
            uint32_t d[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t x = row[4 * fld + k];
                d[2 * k] = (uint32_t)x;
                d[2 * k + 1] = (uint32_t)(x >> 32);
            }
            q[fld] = ent_nz16(d, f.ones16);
        }
        const uint32_t lo = q[0] | (q[1] << 16), hi = q[2] | (q[3] << 16);
        mask = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    int last = ent_prev_last(mask, sb);   // every lane of the wavefront takes part in the exchange

    // place of this lane's symbols in the group's list: one per non-zero value, one more for the run that closes the
    // macroblock (rle.rs:31-38) unless its last coefficient is non-zero
    const int my_last = mask ? 64 * sb + 63 - __builtin_clzll(mask) : last;
    const bool closing = coded && sb == 3 && my_last < 255;
    const uint32_t n_sym = (uint32_t)__popcll(mask) + (closing ? 1u : 0u);
    const uint32_t incl = ent_wave_scan(n_sym);
    if ((threadIdx.x & 63u) == 63u) wave_syms[threadIdx.x >> 6] = incl;
    // symbol counts: 16 byte-wide counters per lane in LDS (ds_add without return: nothing to wait for in the loop)
    cnt4[threadIdx.x] = make_uint4(0, 0, 0, 0);
    ENT_MARK(0, 3);
    __syncthreads();
    ENT_MARK(0, 4);
    uint32_t sym_off = incl - n_sym;
#pragma unroll
    for (unsigned w = 0; w + 1 < kEntThreads / 64; w++) sym_off += w < (threadIdx.x >> 6) ? wave_syms[w] : 0u;
    uint32_t *out = b.syms + ((size_t)stream * f.n_groups + blockIdx.x) * kEntGroupSyms + sym_off;

    uint32_t *my_cnt = (uint32_t *)&cnt4[threadIdx.x];
    auto count = [&](unsigned bin, unsigned n) { atomicAdd(&my_cnt[bin >> 2], n << (8u * (bin & 3u))); };
    const int16_t *c = (const int16_t *)row;
    uint32_t sumsize = 0, maxsize = 0;
    if (mask) {
        uint64_t mm = mask;
        int bit = __builtin_ctzll(mm);
        int v = c[bit];
        for (;;) {
            mm &= mm - 1;
            const int nbit = mm ? __builtin_ctzll(mm) : 0;
            const int nv = c[nbit];   // next value on its way while this one is processed
            const int i = 64 * sb + bit;
            unsigned run = (unsigned)(i - last - 1), fillers = 0;
            last = i;
            if (run > 15u) {
                ent_split_run(run, fillers, run);
                count(15u, fillers);
                count(0u, fillers);
            }
            const unsigned mag = (unsigned)(v < 0 ? -v : v);
            const unsigned size = 33u - (unsigned)__builtin_clz(mag);   // bit length + 1 (rle.rs:23-24)
            maxsize = max(maxsize, size);
            count(run, 1u);
            count(size & 15u, 1u);
            sumsize += size;
            // write_signed keeps the low `size` bits of the two's complement (enc.rs:313-315)
            *out++ = (fillers << 24) | (((uint32_t)v & ((1u << (size & 15u)) - 1u)) << 8) | ((size & 15u) << 4) | run;
            if (!mm) break;
            bit = nbit;
            v = nv;
        }
    }
    if (closing) {   // the trailing run closes the macroblock (rle.rs:31-38): (rest, size 0) after its fillers
        unsigned fillers, rest;
        ent_split_run((unsigned)(255 - last), fillers, rest);
        count(15u, fillers);
        count(0u, fillers + 1u);
        count(rest, 1u);
        *out++ = (fillers << 24) | rest;
    }
    ENT_MARK(0, 5);
    uint32_t hdr_bits = 0;
    if (f.pframe && live && sb == 0)      // has_mvec, has_coeff, then two 7-bit components (enc.rs:414-451)
        hdr_bits = (mvx != 0 || mvy != 0) ? 16u : 2u;
    ent_wave_lds_sync();
    const uint4 c4 = cnt4[threadIdx.x];

    // workgroup totals: counts widened to 16-bit fields (two symbols per word; at most 256 * 164 per field), summed over each
    // 16-lane row with DPP; the 16 row totals go to LDS with plain stores and are added up after the barrier (LDS atomics on
    // one address from four lanes of a wavefront are turned into a scalar loop over the lanes by the compiler -- ten of
    // those cost more than the sums themselves)
    const uint32_t w8[4] = {c4.x, c4.y, c4.z, c4.w};
    const bool lead = (threadIdx.x & 15u) == 0;
    uint32_t *my_part = part[threadIdx.x >> 4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t a = ent_row_sum((w8[j] & 0xffu) | ((w8[j] & 0xff00u) << 8));
        const uint32_t c2 = ent_row_sum(((w8[j] >> 16) & 0xffu) | ((w8[j] >> 24) << 16));
        if (lead) { my_part[2 * j] = a; my_part[2 * j + 1] = c2; }
    }
    const uint32_t ws = ent_row_sum(sumsize), wh = ent_row_sum(hdr_bits);
    if (lead) { my_part[8] = ws; my_part[9] = wh; }
    if (__any(maxsize > 15u) && (threadIdx.x & 63u) == 0) atomicOr(&b.codes[stream].oversize, 1u);
    ENT_MARK(0, 6);
    __syncthreads();
    ENT_MARK(0, 7);
    if (threadIdx.x < 32) {   // threads 0..9: total k = thread; threads 16..31: symbol (thread - 16) of the frame histogram
        const unsigned sym = threadIdx.x - 16u, k = threadIdx.x < 16 ? min(threadIdx.x, 9u) : sym >> 1;
        uint32_t total = 0;
#pragma unroll
        for (int r = 0; r < kEntThreads / 16; r++) total += part[r][k];
        EntGroup *g = b.groups + (size_t)stream * f.n_groups + blockIdx.x;
        if (threadIdx.x < 8) g->counts[threadIdx.x] = total;
        if (threadIdx.x == 8) g->sumsize = total;
        if (threadIdx.x == 9) g->hdr_bits = total;
        if (threadIdx.x == 10) g->n_syms = wave_syms[0] + wave_syms[1] + wave_syms[2] + wave_syms[3];
        if (threadIdx.x >= 16) {
            const uint32_t n = (sym & 1u) ? (total >> 16) : (total & 0xffffu);
            if (n) atomicAdd(&b.hist[stream * 16 + (int)sym], (int32_t)n);
        }
    }
    ENT_MARK(0, 8);
}

// ---------------------------------------------------------------------------------------------------- k_ent_codes
// HuffmanTree::from_table (huffman.rs:71-119) + assign_codes (:204-217), lane ch = symbol ch, one wavefront:
//  * leaves in symbol order for the non-zero table entries, list = stable sort by descending frequency (:81)
//  * while more than one entry: pop the last two (a = last, b = the one before), new node {left a, right b} with the
//    summed frequency goes in front of the first strictly smaller entry (:61-69, :84-93)
//  * code of a leaf: branches from the root down, first branch in bit 0, left = 0, right = 1 (:204-217)
// Same construction, tie-breaks included, as HuffmanTree in pfv_host.hip.
__device__ inline void ent_build_codes_wave(const int32_t *hist_in, uint8_t *table_out, uint32_t *val_out, uint8_t *len_out,
                                            int *parent /*[32] LDS*/, int *branch /*[32] LDS*/)
{
    const int lane = (int)(threadIdx.x & 63u);
    const int32_t h = lane < 16 ? hist_in[lane] : 0;
    int32_t mx = h;
    for (int m = 1; m < 16; m <<= 1) {
        const int32_t o = __shfl_xor(mx, m);
        mx = o > mx ? o : mx;
    }
    uint32_t t = 0;   // rle_create_huffman (rle.rs:49-66)
    if (h > 0) {
        t = (uint32_t)(((uint64_t)(uint32_t)h * 255u) / (uint32_t)mx);
        t = t < 1u ? 1u : t;
    }
    const bool present = lane < 16 && t != 0;
    const uint64_t pm = __ballot(present);
    int n = __popcll(pm), n_nodes = n;
    const int leaf = __popcll(pm & ((1ull << lane) - 1ull));
    // position in the stably sorted list
    int pos = 0;
    for (int j = 0; j < 16; j++) {
        const uint32_t tj = ent_shfl(t, j);
        if (((pm >> j) & 1ull) && (tj > t || (tj == t && j < lane))) pos++;
    }
    if (lane < 32) { parent[lane] = -1; branch[lane] = 0; }
    // list entry p lives on lane p: scatter through LDS (parent[] doubles as scratch before the merges start)
    int *scratch_f = parent, *scratch_n = branch;
    ent_wave_lds_sync();
    if (present) { scratch_f[pos] = (int)t; scratch_n[pos] = leaf; }
    ent_wave_lds_sync();
    uint32_t lf = lane < n ? (uint32_t)scratch_f[lane] : 0u;
    int ln = lane < n ? scratch_n[lane] : -1;
    ent_wave_lds_sync();
    if (lane < 32) { parent[lane] = -1; branch[lane] = 0; }
    ent_wave_lds_sync();
    while (n > 1) {
        const int a = (int)ent_shfl((uint32_t)ln, n - 1), bn = (int)ent_shfl((uint32_t)ln, n - 2);
        const uint32_t fq = ent_shfl(lf, n - 1) + ent_shfl(lf, n - 2);
        n -= 2;
        if (lane == 0) {
            parent[a] = n_nodes; branch[a] = 0;    // left
            parent[bn] = n_nodes; branch[bn] = 1;  // right
        }
        const int ins = __popcll(__ballot(lane < n && !(fq > lf)));   // entries that stay in front of the new node
        const uint32_t up_f = ent_shfl(lf, lane > 0 ? lane - 1 : 0);
        const int up_n = (int)ent_shfl((uint32_t)ln, lane > 0 ? lane - 1 : 0);
        if (lane == ins) { lf = fq; ln = n_nodes; }
        else if (lane > ins && lane <= n) { lf = up_f; ln = up_n; }
        n++;
        n_nodes++;
    }
    ent_wave_lds_sync();
    uint32_t code = 0;
    uint32_t len = 0;
    if (present) {
        for (int nd = leaf; parent[nd] >= 0; nd = parent[nd]) {
            code = (code << 1) | (uint32_t)branch[nd];
            len++;
        }
    }
    if (lane < 16) {
        table_out[lane] = (uint8_t)t;
        val_out[lane] = code;
        len_out[lane] = (uint8_t)len;
    }
}

__global__ void __launch_bounds__(kEntThreads) k_ent_codes(EntFrame f, EntBufs b)
{
    __shared__ uint32_t val[16];
    __shared__ uint8_t len[16], table[16];
    __shared__ int parent[32], branch[32];
    __shared__ uint32_t wave_tot[2][kEntThreads / 64];
    const int stream = (int)blockIdx.x;
    EntCodes *out = b.codes + stream;
    if (threadIdx.x < 64) {
        ent_build_codes_wave(b.hist + stream * 16, table, val, len, parent, branch);
        if (threadIdx.x < 16) b.hist[stream * 16 + threadIdx.x] = 0;   // ready for the next frame
    }
    __syncthreads();
    const unsigned z = threadIdx.x & 15u, nb = threadIdx.x >> 4;
    out->pair_bits[threadIdx.x] = val[z] | (val[nb] << len[z]);
    out->pair_len[threadIdx.x] = (uint8_t)(len[z] + len[nb]);
    if (threadIdx.x < 16) {
        out->len[threadIdx.x] = len[threadIdx.x];
        out->table[threadIdx.x] = table[threadIdx.x];
    }
    // base bit offsets of the scan workgroups: all block headers first (p-frames), then the symbols (enc.rs:414-466)
    EntGroup *groups = b.groups + (size_t)stream * f.n_groups;
    uint32_t hdr_run = 19u * 8u;   // 16 table bytes + 3 q-table indices (enc.rs:290-298, :403-411)
    uint32_t sym_run = 0;          // relative to the end of the header section until the second pass
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: totals of the header section (and the symbol offsets relative to its end); pass 1 writes the bases
        uint32_t h_run = hdr_run, s_run = pass == 0 ? 0u : sym_run;
        for (int g0 = 0; g0 < f.n_groups; g0 += kEntThreads) {
            const int gi = g0 + (int)threadIdx.x;
            uint32_t hb = 0, sbits = 0;
            if (gi < f.n_groups) {
                const EntGroup &g = groups[gi];
                hb = g.hdr_bits;
                sbits = g.sumsize;
#pragma unroll
                for (int k = 0; k < 16; k++) sbits += ((g.counts[k >> 1] >> (16 * (k & 1))) & 0xffffu) * len[k];
            }
            const uint32_t hi = ent_wave_scan(hb), si = ent_wave_scan(sbits);
            __syncthreads();
            if (lane == 63) { wave_tot[0][wave] = hi; wave_tot[1][wave] = si; }
            __syncthreads();
            uint32_t h_before = 0, s_before = 0, h_all = 0, s_all = 0;
            for (int w = 0; w < kEntThreads / 64; w++) {
                if (w < wave) { h_before += wave_tot[0][w]; s_before += wave_tot[1][w]; }
                h_all += wave_tot[0][w];
                s_all += wave_tot[1][w];
            }
            if (pass == 1 && gi < f.n_groups) {
                groups[gi].hdr_base = h_run + h_before + hi - hb;
                groups[gi].sym_base = s_run + s_before + si - sbits;
            }
            h_run += h_all;
            s_run += s_all;
        }
        if (pass == 0) {
            sym_run = h_run;                                  // symbols start where the headers end
            if (threadIdx.x == 0) {
                uint32_t bytes = (h_run + s_run + 7u) >> 3;   // byte_align (enc.rs:318, :468)
                if (out->oversize) bytes = kEntErrOversize;
                else if (bytes > f.cap_bytes) bytes = kEntErrCapacity;
                b.sizes[stream] = bytes;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- k_ent_init
__global__ void __launch_bounds__(kEntThreads) k_ent_init(EntFrame f, EntBufs b)
{
    const int stream = (int)blockIdx.y;
    const uint32_t bytes = b.sizes[stream];
    if (blockIdx.x == 0 && threadIdx.x == 0) b.codes[stream].oversize = 0;   // consumed by k_ent_codes; next frame starts clean
    if (bytes >= kEntErrCapacity) return;
    uint32_t *w = (uint32_t *)(b.payload + (size_t)stream * f.cap_bytes);
    const uint32_t n_words = (bytes + 3u) >> 2;
    const uint8_t *t = b.codes[stream].table;
    for (uint32_t i = blockIdx.x * kEntThreads + threadIdx.x; i < n_words; i += gridDim.x * kEntThreads) {
        uint32_t v = 0;
        if (i < 4) v = (uint32_t)t[4 * i] | ((uint32_t)t[4 * i + 1] << 8) | ((uint32_t)t[4 * i + 2] << 16) | ((uint32_t)t[4 * i + 3] << 24);
        else if (i == 4) v = (uint32_t)f.qidx[0] | ((uint32_t)f.qidx[1] << 8) | ((uint32_t)f.qidx[2] << 16);
        w[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------- k_ent_pack
// A workgroup owns one contiguous bit range of the symbol section (and, for p-frames, one of the header section).  Its
// lanes OR their bits into an LDS window anchored at the range's first word; the window leaves with coalesced dword
// stores, only its first and last word -- shared with the neighbouring workgroups -- as atomicOr onto the zeroed
// payload.  The window holds kEntWinWords * 32 bits (~6x the typical group at quality 5); a denser group sends the
// finished words out whenever the next 256 symbols would not fit and re-anchors the window at the running offset.
#ifndef PFV_ENT_WIN_WORDS
#define PFV_ENT_WIN_WORDS 2048   // tests build a 64-word variant so that small frames reach the re-anchoring and memory paths
#endif
constexpr int kEntWinWords = PFV_ENT_WIN_WORDS;
constexpr int kEntHdrWords = 64 * 16 / 32 + 2;   // 64 macroblocks x 16 header bits, plus unaligned ends

// Bits at bit `off` of an LDS window (ds_or_b32, nothing returned): no carried accumulator and no branches -- ORing
// zero bits is harmless, so every call touches all the words the longest case can reach.  ent_or_bits: bits < 2^45
// (a code pair of at most 30 bits + 15 value bits) -> three words; ent_or_bits32: bits < 2^32 -> two words.
__device__ __forceinline__ void ent_or_bits32(uint32_t *win, uint32_t off, uint32_t bits)
{
    const uint32_t sh = off & 31u;
    uint32_t *w = win + (off >> 5);
    const uint64_t x = (uint64_t)bits << sh;
    atomicOr(w, (uint32_t)x);
    atomicOr(w + 1, (uint32_t)(x >> 32));
}
__device__ __forceinline__ void ent_or_bits(uint32_t *win, uint32_t off, uint64_t bits)
{
    const uint32_t sh = off & 31u;
    uint32_t *w = win + (off >> 5);
    const uint64_t x = bits << sh;
    atomicOr(w, (uint32_t)x);
    atomicOr(w + 1, (uint32_t)(x >> 32));
    atomicOr(w + 2, ((uint32_t)(bits >> 32) >> 1) >> (31u - sh));   // bits >> (64 - sh), 0 for sh == 0
}

// The same onto the zeroed payload in memory (a symbol that does not fit the window even after re-anchoring): every word
// may be shared with a neighbour, so every non-zero word is an atomicOr.
__device__ __forceinline__ void ent_or_bits_mem(uint32_t *words, uint32_t off, uint64_t bits)
{
    const uint32_t sh = off & 31u;
    uint32_t *w = words + (off >> 5);
    const uint64_t x = bits << sh;
    const uint32_t x2 = ((uint32_t)(bits >> 32) >> 1) >> (31u - sh);
    if ((uint32_t)x) atomicOr(w, (uint32_t)x);
    if ((uint32_t)(x >> 32)) atomicOr(w + 1, (uint32_t)(x >> 32));
    if (x2) atomicOr(w + 2, x2);
}

// ---------------------------------------------------------------------------------------------------- k_ent_pack
// One workgroup per group of k_ent_scan; the group's symbol list is walked 256 symbols at a time, lane = one symbol
// (not one subblock: the lanes of a subblock-per-lane walk wait for the fullest subblock of the wavefront, 9-10 symbols
// against a mean of 2-3 in a typical p-frame).  Per step: the symbol's code pair and bit length from the LDS tables,
// workgroup prefix sum of the lengths (DPP scan + the four wavefront totals) on top of the running offset, then the bits
// are ORed into the LDS window (enc.rs:307-316, :459-466: fillers, code pair, value bits).  A step that would run past
// the window first sends the finished words out and re-anchors the window at the running offset.
struct EntWindow {
    uint32_t *win;        // LDS, kEntWinWords + 2 words
    uint32_t *words;      // the stream's payload
    uint32_t word0;       // payload word of win[0]
    // Words [0, end_bit / 32) go out, the first as atomicOr (shared with the previous group or carried over from the
    // previous anchor), the others as plain stores; the partial last word follows as atomicOr when `close`, else it is
    // returned to be carried into the next anchor.  Workgroup-uniform arguments; the caller has a barrier on either side.
    __device__ __forceinline__ uint32_t flush(uint32_t end_bit, bool close) const
    {
        const uint32_t n_full = end_bit >> 5;
        for (uint32_t i = threadIdx.x; i < n_full; i += kEntThreads) {
            const uint32_t x = win[i];
            if (i == 0) { if (x) atomicOr(&words[word0], x); }
            else words[word0 + i] = x;
        }
        if (!(end_bit & 31u)) return 0;
        const uint32_t x = win[n_full];
        if (close) {
            if (threadIdx.x == 0 && x) atomicOr(&words[word0 + n_full], x);
            return 0;
        }
        return x;
    }
};

__global__ void __launch_bounds__(kEntThreads) k_ent_pack(EntFrame f, EntBufs b)
{
    __shared__ uint32_t win[kEntWinWords + 2];   // + the two words ent_or_bits may touch past a symbol's last bit
    __shared__ uint32_t hwin[kEntHdrWords + 1];
    __shared__ uint32_t pair_bits[256];
    __shared__ uint8_t pair_len[256];
    __shared__ uint32_t wave_tot[2][kEntThreads / 64];
    __shared__ uint32_t cut;                     // window-relative bit where the symbols that went to memory start
    __shared__ uint32_t hdr_total;
    constexpr uint32_t kWinBits = (uint32_t)kEntWinWords * 32u, kNoCut = 0xffffffffu;
    const int stream = (int)blockIdx.y;
    ENT_MARK0();
    ENT_MARK(1, 0);
    // every global read of the kernel's head is issued here, before anything waits
    const uint32_t stream_bytes = b.sizes[stream];
    const EntGroup *g = b.groups + (size_t)stream * f.n_groups + blockIdx.x;
    const uint32_t sym_base = g->sym_base, hdr_base = g->hdr_base, n_list = g->n_syms;
    const uint32_t *gsyms = b.syms + ((size_t)stream * f.n_groups + blockIdx.x) * kEntGroupSyms;
    uint32_t w_next = gsyms[threadIdx.x];        // the list's first 256 words exist whatever n_list is
    const EntCodes *codes = b.codes + stream;
    const uint32_t my_pair_bits = codes->pair_bits[threadIdx.x];
    const uint8_t my_pair_len = codes->pair_len[threadIdx.x];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    // block headers (p-frames, enc.rs:414-451): the group's 64 macroblocks on wavefront 0
    const int mbi = (int)blockIdx.x * (kEntThreads / 4) + lane;
    uint32_t my_hdr = 0, hdr_bits = 0;
    if (f.pframe && wave == 0 && mbi < f.total_blocks) {
        const size_t bi = (size_t)stream * f.total_blocks + mbi;
        const int mvx = b.mv[2 * bi], mvy = b.mv[2 * bi + 1];
        const bool has_coef = b.has[bi] != 0;
        my_hdr = (mvx != 0 || mvy != 0) ? 16u : 2u;
        hdr_bits = (my_hdr == 16u ? 1u : 0u) | (has_coef ? 2u : 0u);
        if (my_hdr == 16u) hdr_bits |= ((uint32_t)mvx & 0x7fu) << 2 | ((uint32_t)mvy & 0x7fu) << 9;
    }
    if (stream_bytes >= kEntErrCapacity) return;   // uniform over the workgroup

    pair_bits[threadIdx.x] = my_pair_bits;
    pair_len[threadIdx.x] = my_pair_len;
    for (int i = (int)threadIdx.x; i < kEntWinWords + 2; i += kEntThreads) win[i] = 0;
    if (threadIdx.x <= kEntHdrWords) hwin[threadIdx.x] = 0;
    if (threadIdx.x == 0) cut = kNoCut;
    ENT_MARK(1, 1);
    __syncthreads();
    ENT_MARK(1, 2);
    uint32_t *words = (uint32_t *)(b.payload + (size_t)stream * f.cap_bytes);
    const uint32_t hword0 = hdr_base >> 5;
    if (f.pframe && wave == 0) {
        const uint32_t hi = ent_wave_scan(my_hdr);
        if (my_hdr) ent_or_bits32(hwin, (hdr_base & 31u) + hi - my_hdr, hdr_bits);
        if (lane == 63) hdr_total = hi;
    }

    const uint32_t filler_bits = pair_bits[15], filler_len = pair_len[15];   // (15, size 0)
    EntWindow wnd{win, words, sym_base >> 5};
    uint32_t running = sym_base & 31u;           // window-relative bit of the next symbol (workgroup-uniform)
    int parity = 0;
    for (uint32_t base = 0; base < n_list; base += kEntThreads, parity ^= 1) {
        const uint32_t j = base + threadIdx.x;
        const bool active = j < n_list;
        const uint32_t w = active ? w_next : 0u;
        if (j + kEntThreads < n_list) w_next = gsyms[j + kEntThreads];   // the next step's word on its way
        const uint32_t fillers = w >> 24, p = w & 255u, size = (w >> 4) & 15u;
        const uint32_t pb = pair_bits[p], pl = pair_len[p], vb = (w >> 8) & 0x7fffu;
        const uint32_t len = active ? fillers * filler_len + pl + size : 0u;
        const uint32_t incl = ent_wave_scan(len);
        if (lane == 63) wave_tot[parity][wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kEntThreads / 64; k++) {
            const uint32_t t = wave_tot[parity][k];
            if (k < wave) before += t;
            total += t;
        }
        if (running + total > kWinBits) {        // uniform: send the finished words out, anchor the window at `running`
            const uint32_t stop = min(running, cut);
            const uint32_t carry = wnd.flush(stop, stop != running);
            __syncthreads();
            for (int i = (int)threadIdx.x; i < kEntWinWords + 2; i += kEntThreads) win[i] = i ? 0u : carry;
            if (threadIdx.x == 0) cut = kNoCut;
            wnd.word0 += running >> 5;
            running &= 31u;
            __syncthreads();
        }
        uint32_t off = running + before + incl - len;
        if (active) {
            const uint64_t code = (uint64_t)pb | ((uint64_t)vb << pl);
            if (off + len <= kWinBits) {
                for (uint32_t fl = fillers; fl; fl--) {
                    ent_or_bits32(win, off, filler_bits);
                    off += filler_len;
                }
                ent_or_bits(win, off, code);
            } else {                             // a single step larger than the window: its tail goes straight to memory
                atomicMin(&cut, off);
                uint32_t abs_off = (wnd.word0 << 5) + off;
                for (uint32_t fl = fillers; fl; fl--) {
                    ent_or_bits_mem(words, abs_off, filler_bits);
                    abs_off += filler_len;
                }
                ent_or_bits_mem(words, abs_off, code);
            }
        }
        running += total;
    }
    ENT_MARK(1, 5);
    __syncthreads();
    ENT_MARK(1, 6);
    wnd.flush(min(running, cut), true);
    const EntWindow hw{hwin, words, hword0};
    if (f.pframe) hw.flush((hdr_base & 31u) + hdr_total, true);
    ENT_MARK(1, 7);
}

// ---------------------------------------------------------------------------------------------------- k_ent_gather
// All streams' payloads back to back (each start 16-byte aligned) so that ONE device-to-host copy fetches them.
// offsets[s] = start of stream s in `packed`; sizes as left by k_ent_codes (error markers copy nothing).
__global__ void __launch_bounds__(kEntThreads) k_ent_gather(EntFrame f, EntBufs b, const uint32_t *offsets, uint8_t *packed)
{
    const int stream = (int)blockIdx.y;
    const uint32_t bytes = b.sizes[stream];
    if (bytes >= kEntErrCapacity) return;
    const uint4 *src = (const uint4 *)(b.payload + (size_t)stream * f.cap_bytes);
    uint4 *dst = (uint4 *)(packed + offsets[stream]);
    const uint32_t n = (bytes + 15u) >> 4;   // the tail of the last 16 bytes is payload padding / stale words: harmless
    for (uint32_t i = blockIdx.x * kEntThreads + threadIdx.x; i < n; i += gridDim.x * kEntThreads) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------- retained payloads
// pfv_gop_encoder runs a whole batch of frame steps without a host round trip and collects the batch afterwards, so the payloads
// of every step have to outlive the step: after the pack of a step ONE thread appends the payloads of the step's slots to a device
// arena (16-byte aligned starts) -- entries[slot] = {offset, size}; an error marker, or a payload the arena has no room for, keeps
// the marker as its size and nothing is copied -- and k_ent_gather_entries moves the bytes there.
struct EntEntry {
    unsigned long long offset;
    uint32_t size;             // payload bytes, kEntErrOversize or kEntErrCapacity
    uint32_t reserved;
};
// level_host: where the arena's fill level after this launch is also left for the host, in page-locked memory (a store across PCIe instead
// of an 8-byte hipMemcpyAsync per step on the kernel stream: the host reads it once the step's event has fired)
__global__ void k_ent_retain(const uint32_t *sizes, int count, unsigned long long *cursor, unsigned long long cap, EntEntry *entries, volatile unsigned long long *level_host)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long cur = *cursor;
    for (int k = 0; k < count; k++) {
        EntEntry e{cur, sizes[k], 0u};
        if (e.size < kEntErrCapacity) {
            const unsigned long long need = ((unsigned long long)e.size + 15ull) & ~15ull;
            if (cur + need > cap) e.size = kEntErrCapacity;
            else cur += need;
        }
        entries[k] = e;
    }
    *cursor = cur;
    *level_host = cur;
}
__global__ void __launch_bounds__(kEntThreads) k_ent_gather_entries(EntFrame f, EntBufs b, const EntEntry *entries, uint8_t *arena)
{
    const int stream = (int)blockIdx.y;
    const EntEntry e = entries[stream];
    if (e.size >= kEntErrCapacity) return;
    const uint4 *src = (const uint4 *)(b.payload + (size_t)stream * f.cap_bytes);
    uint4 *dst = (uint4 *)(arena + e.offset);
    const uint32_t n = (e.size + 15u) >> 4;
    for (uint32_t i = blockIdx.x * kEntThreads + threadIdx.x; i < n; i += gridDim.x * kEntThreads) dst[i] = src[i];
}

}  // namespace pfv
