// pfv_entdec_kernels.hip -- the decoder's entropy stage on the device (gfx950): packet payloads -> coefficient lists (pfv_device.h: CoefLists).
//
// The reference reads a payload one bit field at a time on one host thread (src/dec.rs:258-296 i-frames: ONE run stream per
// frame; :378-417 p-frames: one run stream per coded macroblock, back to back; per run a (num_zeroes, coeff_size) pair of tree
// codes, src/huffman.rs:156-197, and coeff_size raw bits).  Nothing in the payload says where a macroblock starts, so the
// stream cannot simply be cut up -- but it SELF-SYNCHRONISES: a reader that starts at a wrong bit lands on a true run boundary
// after a handful of runs and is identical to the true reader from there on.  Hence (after Weissenberger & Schmidt's parallel
// Huffman decoding):
//
//   k_entd_sync   the payload behind the block headers is cut into subsequences of kEdSubBits bits, one lane each.  end[i] = the
//                 first run boundary at or behind the end of subsequence i when reading from `start` -- round 1 takes the
//                 subsequence's own first bit for `start` (a guess), every later round takes end[i - 1] and only runs where that
//                 differs from what the lane used before.  Rounds are cheap inside a workgroup (256 lanes pass their ends along in
//                 LDS until nothing changes; the lanes that still have work are packed into as few wavefronts as they need).
//   k_entd_fix    between workgroups the ends travel through memory: one thread per seam reads the workgroup's first lane from its
//                 true start and follows the change until it meets the recorded read.
//   k_entd_verify reads nothing and checks that no lane has anything left to do: then end[0] is true (the first lane starts at the
//                 true first run) and every end[i] follows from a true start.  A packet that has not settled (periodic content can
//                 keep a wrong phase for ever) is left to the host parser.
//   k_entd_prefix exclusive prefix over the coefficients each workgroup's subsequences cover and over the values among them (summed by
//                 the verifying launch): with a workgroup-local prefix in k_entd_emit, the coefficient index every lane's first run
//                 starts at and the place of its first value in the packet's list.
//   k_entd_emit   every lane reads its subsequence once more, from its true start, and appends its values to the packet's coefficient
//                 list: one 32-bit entry per value, contiguous per lane, per workgroup and per packet -- nothing is cleared and nothing
//                 but the values is written (round 4 scattered 2-byte values into a zeroed [macroblock][256] array: 46 x the bytes).
//                 The run that enters a macroblock knows how many values lie before it: the frame's exclusive counts (a p-frame's
//                 through the list of coded macroblocks).  Entries and counts leave the workgroup through LDS as whole lines.
//                 It also decides whether the host parser would have accepted the
//                 payload and produced the same array: anything it is not sure of -- a field that runs past the payload, a value
//                 behind the last coefficient, a macroblock whose runs do not end exactly on its 256th coefficient -- marks the
//                 packet, and a marked packet is parsed by the host code instead (pfv_host.hip: read_runs), which alone
//                 decides about errors.  The device path therefore never has to reproduce an error case.
//
// Codes are looked up in a 12-bit table (LDS, built by the workgroup from the packet's 16 codes), longer codes go through the 16 codes
// one by one; the payload bits a workgroup reads are staged in LDS.  Included by pfv_capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pfv_device.h"

namespace pfv {

constexpr int kEdThreads = 256;
// A workgroup OWNS kEdOwn consecutive subsequences of its packet; k_entd_sync reads kEdHalo more in front of them with its first threads:
// what the workgroup in front computes for its last lanes, again, from a guess -- by the halo's end that read has met the true one (it
// has not with probability ~0.32 per lane: 1e-8 for 16), so the workgroup's first own lane starts right and the seams between workgroups
// need no second pass (k_entd_fix stays as a cheap check; k_entd_verify finds what it cannot repair).
constexpr int kEdHalo = 16;
constexpr int kEdOwn = kEdThreads - kEdHalo;
constexpr uint32_t kEdSubBits = 256;                // payload bits per lane (default; EdPacket::sub_bits, a multiple of 32 up to kEdMaxSubBits, is what the kernels use)
constexpr uint32_t kEdIrregular = 1u;               // k_entd_emit: the host parser decides about this packet
constexpr uint32_t kEdUnsettled = 2u;               // k_entd_sync: the subsequence starts had not settled
constexpr uint32_t kEdNoStart = 0xffffffffu;
constexpr int kEdInner = 96;                        // k_entd_sync: settling rounds inside a workgroup at most (default; a round without work ends them)

// one packet of the batch (made by the host from the packet's first 19 bytes and, p-frames, its block headers)
struct EdPacket {
    unsigned long long byte_off;   // payload position in the device byte buffer (multiple of 4; >= 16 readable bytes behind the payload)
    uint32_t total_bits;           // payload size in bits
    uint32_t bit0;                 // first bit of the run streams (behind the table, the q indices and the block headers); a p-frame whose block
                                   // headers are read on the device (k_hdr_*): written there, with total_coefs, list_cap and first_sub
    uint32_t org;                  // bit the subsequences are counted from: subsequence i is [org + i sub_bits, org + (i + 1) sub_bits).  = bit0 when the
                                   // host read the headers; the first bit behind the q indices (152) when the device does -- the grids are sized
                                   // before anybody knows where the headers end, the subsequences in front of bit0 simply have no lane
    uint32_t first_sub;            // the subsequence bit0 lies in: its lane starts at bit0 (a true run boundary), the ones before it do nothing
    uint32_t hdr_first, hdr_wgs;   // k_hdr_*: the packet's place in the per-workgroup header arrays, its header workgroups (0: headers read on the host)
    uint32_t total_coefs;          // coefficients the run streams cover: macroblocks x 256 (i-frame), coded macroblocks x 256 (p-frame)
    uint32_t n_sub;                // subsequences = ceil((total_bits - org) / sub_bits); 0: nothing to read
    uint32_t sub_bits;             // payload bits per lane
    uint32_t sub_first;            // index of subsequence 0 in the per-subsequence arrays
    uint32_t grp_first;            // index of its first workgroup in the per-workgroup array
    uint32_t list_cap;             // entries the packet's list has room for (entd_list_cap: what its bits can hold at most)
    uint32_t pframe;               // 1: values go through the coded-macroblock list
    uint32_t total_blocks;         // macroblocks per frame
    unsigned long long frame_off;  // which frame of the has / coded-list / range / list-pointer arrays this packet fills
    uint16_t code_val[16];         // the packet's tree codes, LSB first (src/huffman.rs:204-217)
    uint8_t code_len[16];          // 0: the symbol has no code
};

struct EdBufs {
    const uint8_t *bytes;          // payloads
    EdPacket *packets;             // read-only for every kernel but k_hdr_emit, which completes a p-frame's descriptor
    const uint2 *groups;           // workgroups of k_entd_sync / k_entd_emit: (packet, which kEdOwn subsequences of it)
    uint32_t *end, *used, *cnt;    // per subsequence; cnt = coefficients covered | values among them << 16 (a lane reads < 2^9 + 45 bits: < 2^13 of either)
    unsigned long long *wgsum;     // per workgroup of k_entd_sync: the sums of its lanes' cnt fields (coefficients | values << 32), then (k_entd_prefix) those before it
    uint32_t *coded;               // [frame][total_blocks]: the k-th coded macroblock of the frame (p-frames; k_entd_coded, from the has_coeff bytes)
    uint32_t *const *lists;        // [frame]: where the frame's entries go (EdPacket::list_cap of them fit)
    uint32_t *counts;              // [frame][total_blocks + 1]: per macroblock, the entries before it (CoefLists::counts)
    uint32_t *status;              // per packet: kEd* bits
    uint32_t packet0;              // k_entd_prefix: the launch's first packet (one workgroup per packet)
    uint32_t group0;               // index of b.groups[0] among all workgroups of the batch (EdPacket::grp_first counts from there too)
    // block headers read on the device (k_hdr_*; packets with hdr_wgs != 0)
    uint32_t *hdr_maps;            // [header workgroup][8]: where a reader that enters the workgroup's bits at entry e leaves them, and what it passes
    uint4 *hdr_start;              // [header workgroup]: the TRUE reader's entry, macroblocks and coded macroblocks before the workgroup
    int8_t *mv;                    // [frame][total_blocks][2]   written by k_hdr_emit
    uint8_t *has;                  // [frame][total_blocks]
};

// tab[v] = (code length | symbol << 4) of the tree code at the low end of the 12 bits v, 0 when that code is longer than 12 bits.
// One lookup per code and no (num_zeroes, coeff_size) pair table: a pair table needs a second path for the pairs it cannot hold, and with
// 64 lanes per wavefront nearly every step had SOME lane on that path -- the wavefront then pays for both.
__device__ __forceinline__ void ed_build_table(uint8_t *tab, uint16_t *cval, uint8_t *clen, const EdPacket &pk, int tid)
{
    if (tid < 16) { cval[tid] = pk.code_val[tid]; clen[tid] = pk.code_len[tid]; }
    uint32_t *tab32 = (uint32_t *)tab;
    for (int k = tid; k < 1024; k += kEdThreads) tab32[k] = 0u;           // what no code of <= 12 bits claims stays 0
    __syncthreads();
    // a code of l bits owns the 2^(12 - l) entries whose low l bits are the code: 4 096 stores in all (the codes are prefix-free), not
    // 4 096 x 16 comparisons -- the table is rebuilt by every workgroup of every launch that reads, and was half of k_entd_emit's VALU work
    for (uint32_t s = 0; s < 16; s++) {
        const uint32_t l = clen[s];
        if (l == 0 || l > 12) continue;
        const uint32_t e = l | (s << 4), base = cval[s], n = 4096u >> l;
        for (uint32_t k = (uint32_t)tid; k < n; k += kEdThreads) tab[base | (k << l)] = (uint8_t)e;
    }
    __syncthreads();
}

// The bits a workgroup reads -- its 256 subsequences and what the last run of a lane hangs over -- are staged in LDS once (coalesced);
// a lane then takes 32-bit windows at any bit position with one two-word read and one v_alignbit, and keeps no bit-buffer state.
constexpr uint32_t kEdMaxSubBits = 256;                                    // the staging area is sized for this
constexpr uint32_t kEdStageWords = kEdThreads * kEdMaxSubBits / 32 + 8;    // + 256 bits: a run starts < 45 bits behind a lane's limit, a window reads 64 behind its position
struct EdReader {
    const uint32_t *lw;    // LDS: the payload from bit `base` on
    uint32_t base, pos;
    __device__ __forceinline__ uint32_t window() const   // the 32 payload bits from pos on
    {
        const uint32_t rel = pos - base, k = rel >> 5;
        return __builtin_amdgcn_alignbit(lw[k + 1], lw[k], rel & 31u);
    }
};
// stage the bits of kEdThreads lanes from subsequence `first_lane` on: words [first_bit / 32, ...) of the payload; beyond the payload's own words (+ 3: the slack the host left) zeros
__device__ __forceinline__ uint32_t ed_stage(uint32_t *lw, const uint8_t *bytes, const EdPacket &pk, uint32_t first_lane, int tid)
{
    const unsigned long long first_bit = (unsigned long long)pk.org + (unsigned long long)first_lane * pk.sub_bits;
    const uint32_t w0 = (uint32_t)(first_bit >> 5), n = kEdThreads * pk.sub_bits / 32u + 8u, have = (pk.total_bits + 31u) / 32u + 3u;
    const uint32_t *src = (const uint32_t *)(bytes + pk.byte_off);
    for (uint32_t k = (uint32_t)tid; k < n; k += kEdThreads) lw[k] = w0 + k < have ? src[w0 + k] : 0u;
    return w0 * 32u;
}

// a tree code longer than 12 bits, through the 16 codes (a tree of >= 2 symbols is full: exactly one code matches any bit pattern)
__device__ __forceinline__ uint32_t ed_long_code(uint32_t w, const uint16_t *cval, const uint8_t *clen)
{
    uint32_t e = 1u;
    for (uint32_t s = 0; s < 16; s++) {
        const uint32_t l = clen[s];
        if (l && (w & ((1u << l) - 1u)) == cval[s]) e = l | (s << 4);
    }
    return e;
}

// one run: num_zeroes, coeff_size, the value bits (sign-extended); the reader ends up on the next run
__device__ __forceinline__ void ed_run(EdReader &r, const uint8_t *tab, const uint16_t *cval, const uint8_t *clen, uint32_t &zeros, uint32_t &nb, int &value)
{
    uint32_t w = r.window();
    uint32_t e = tab[w & 4095u];
    if (__builtin_expect(e == 0, 0)) e = ed_long_code(w, cval, clen);
    uint32_t used = e & 15u;
    zeros = e >> 4;
    if (__builtin_expect(used > 12, 0)) { r.pos += used; used = 0; w = r.window(); }       // keep >= 20 bits in the window
    else w >>= used;
    e = tab[w & 4095u];
    if (__builtin_expect(e == 0, 0)) e = ed_long_code(w, cval, clen);
    const uint32_t l2 = e & 15u;
    nb = e >> 4;
    used += l2;
    w >>= l2;
    if (__builtin_expect(used + nb > 32, 0)) { r.pos += used; used = 0; w = r.window(); }  // the value's bits are not all in this window
    const uint32_t raw = w & ((1u << nb) - 1u), sign = (1u << nb) >> 1;
    value = (int)((raw ^ sign) - sign);
    r.pos += used + nb;
}

// The same reader with the bits in REGISTERS (round 5): `buf` holds the payload from `pos` on, at least 33 valid bits of it, and the staged
// word behind them is already on its way (`nxt`, fetched a word -- one or two runs -- before it is shifted in).  With EdReader every run
// was three LDS round trips one after the other (window, first code, second code); the settling rounds and the emit pass are nothing but such
// chains, a dozen runs per lane and ~21 rounds per launch.  Here the window costs no trip, and the two codes cost ONE where they fit 12 bits
// together (ed_build_pairs): one dependent LDS read per run.  The pair table's miss path (two single lookups) is taken by a whole wavefront
// as soon as one lane misses -- which made round 4 drop a pair table: nearly every step of 64 lanes had a miss -- but the rounds that make up the
// kernels' time run a handful of packed lanes, and a handful rarely misses.
struct EdBitBuf {
    const uint32_t *lw;        // LDS: the payload from bit `base` on
    uint32_t base, pos;
    unsigned long long buf;    // bits pos .. pos + have - 1, low bit first
    uint32_t have, kw, nxt;    // valid bits in buf (> 32); lw[kw] = nxt is the word that follows them
    __device__ __forceinline__ void start(const uint32_t *words, uint32_t base_bit, uint32_t at)
    {
        lw = words; base = base_bit; pos = at;
        const uint32_t rel = at - base_bit, k = rel >> 5, sh = rel & 31u;
        buf = (((unsigned long long)lw[k + 1] << 32) | lw[k]) >> sh;
        have = 64u - sh;
        kw = k + 2u;
        nxt = lw[kw];
    }
    __device__ __forceinline__ uint32_t window() const { return (uint32_t)buf; }
    __device__ __forceinline__ void consume(uint32_t n)      // n <= 15
    {
        buf >>= n; have -= n; pos += n;
        if (have <= 32u) {
            buf |= (unsigned long long)nxt << have;
            have += 32u;
            kw++;
            nxt = lw[kw];
        }
    }
};
// ptab[v] = 0x8000 | coeff_size << 8 | num_zeroes << 4 | bits of both codes, where the two tree codes at the low end of the 12 bits v take 12 bits
// or fewer together; 0 otherwise.  From the single-code table: 16 entries per thread.
__device__ __forceinline__ void ed_build_pairs(uint16_t *ptab, const uint8_t *tab, int tid)
{
    for (uint32_t v = (uint32_t)tid; v < 4096u; v += kEdThreads) {
        uint32_t p = 0;
        const uint32_t e1 = tab[v];
        if (e1) {
            const uint32_t l1 = e1 & 15u, e2 = tab[v >> l1], l2 = e2 & 15u;     // the bits behind the first code, zero-filled at the top: a second code of
            if (e2 && l1 + l2 <= 12u) p = 0x8000u | ((e2 >> 4) << 8) | (e1 & 0xf0u) | (l1 + l2);   // l2 <= 12 - l1 bits lies wholly inside what is known
        }
        ptab[v] = (uint16_t)p;
    }
    __syncthreads();
}
// PAIRS: try the pair table first.  Measured per ten 4K packets (profiles/r05_entropy_decoder.md): k_entd_sync 82 us with EdReader, 86 with this
// reader and single lookups, 75 with the pair table in every round (its time is the ~20 late rounds of a few packed lanes, where a miss is rare);
// k_entd_emit -- one pass of full wavefronts, some lane misses in nearly every step and the wavefront pays for both paths -- 90 us with EdReader,
// 95 with this reader, 106 with the pair table: it keeps EdReader.
template <bool PAIRS>
__device__ __forceinline__ void ed_run(EdBitBuf &r, const uint16_t *ptab, const uint8_t *tab, const uint16_t *cval, const uint8_t *clen, uint32_t &zeros, uint32_t &nb,
                                       int &value)
{
    const uint32_t pe = PAIRS ? ptab[r.window() & 4095u] : 0u;
    if (PAIRS && __builtin_expect((pe & 0x8000u) != 0, 1)) {
        zeros = (pe >> 4) & 15u;
        nb = (pe >> 8) & 15u;
        r.consume(pe & 15u);
    } else {
        uint32_t w = r.window();
        uint32_t e = tab[w & 4095u];
        if (__builtin_expect(e == 0, 0)) e = ed_long_code(w, cval, clen);
        zeros = e >> 4;
        r.consume(e & 15u);
        w = r.window();
        e = tab[w & 4095u];
        if (__builtin_expect(e == 0, 0)) e = ed_long_code(w, cval, clen);
        nb = e >> 4;
        r.consume(e & 15u);
    }
    const uint32_t raw = r.window() & ((1u << nb) - 1u), sign = (1u << nb) >> 1;
    value = (int)((raw ^ sign) - sign);
    r.consume(nb);
}

__device__ __forceinline__ uint32_t ed_limit(const EdPacket &pk, uint32_t i)
{
    const unsigned long long lim = (unsigned long long)pk.org + (unsigned long long)(i + 1u) * pk.sub_bits;
    return lim < pk.total_bits ? (uint32_t)lim : pk.total_bits;
}

// workgroup scan helper: exclusive prefix of one value per lane (kEdThreads lanes), total in *sum.  The values carry two counters
// (coefficients | values << 32) that cannot carry into each other: a packet's run streams are < 2^32 bits and a coefficient costs a bit
// or more only for degenerate tables, which never come here.
__device__ __forceinline__ unsigned long long ed_block_exclusive(unsigned long long v, unsigned long long *scratch, int tid, unsigned long long *sum)
{
    scratch[tid] = v;
    __syncthreads();
    for (int d = 1; d < kEdThreads; d <<= 1) {
        const unsigned long long add = tid >= d ? scratch[tid - d] : 0ull;
        __syncthreads();
        scratch[tid] += add;
        __syncthreads();
    }
    const unsigned long long incl = scratch[tid];
    if (sum) *sum = scratch[kEdThreads - 1];
    __syncthreads();
    return incl - v;
}
__device__ __forceinline__ unsigned long long ed_split(uint32_t cnt) { return (unsigned long long)(cnt & 0xffffu) | ((unsigned long long)(cnt >> 16) << 32); }

// one workgroup per entry of b.groups: every lane reads its subsequence from the guess, then the lanes pass their ends along through LDS
// and those with a new start read again, until none has one (a lane whose read had not met the true one by its end changes its
// neighbour's start, and so on: ~68 % of the lanes are right after the second read, a third of the rest after each further one).  From the
// third round on few lanes have work, but a wavefront with ONE such lane takes as long as a full one: the lanes with work are packed
// (ballot + popcount ranks -> a list in LDS) and thread t reads for the t-th of them, so a round occupies ceil(n / 64) wavefronts, not 4.
// (Measured and dropped: letting a thread run on into the lanes behind its own while its read ends elsewhere than the recorded one -- the
// correction then crosses a wrong-phase stretch in one round instead of one lane per round -- made the whole decode 24 % SLOWER, 0.96
// against 1.26 G macroblocks/s for config 4 with one entropy stream: p-frames hold a wrong phase for 20 lanes and more, and such a
// stretch read by ONE thread idles the other 63 lanes of its wavefront for its whole length.)
// The workgroup's first lane keeps its guess: it is the first of the halo (see kEdHalo), whose lanes are not written back.
__global__ void __launch_bounds__(kEdThreads) k_entd_sync(EdBufs b, int inner)
{
    __shared__ __attribute__((aligned(16))) uint8_t tab[4096];
    __shared__ uint16_t cval[16];
    __shared__ uint8_t clen[16];
    __shared__ uint32_t lw[kEdStageWords];
    __shared__ uint16_t ptab[4096];
    __shared__ uint32_t s_used[kEdThreads], s_end[kEdThreads], s_cnt[kEdThreads];     // the lanes' state
    __shared__ uint32_t s_list[kEdThreads], s_start[kEdThreads];                        // this round's lanes with work, packed
    __shared__ uint32_t s_wt[kEdThreads / 64];
    const uint2 grp = b.groups[blockIdx.x];
    const EdPacket &pk = b.packets[grp.x];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // thread t reads subsequence i0 + t: the halo (t < kEdHalo, none in a packet's first workgroup), then the workgroup's own
    const uint32_t halo = grp.y ? (uint32_t)kEdHalo : 0u, i0 = grp.y * kEdOwn - halo, i = i0 + (uint32_t)tid;
    const bool mine = (uint32_t)tid < halo + kEdOwn && i < pk.n_sub && i >= pk.first_sub;
    const uint32_t base = ed_stage(lw, b.bytes, pk, i0, tid);
    s_used[tid] = kEdNoStart; s_end[tid] = 0; s_cnt[tid] = 0;
    ed_build_table(tab, cval, clen, pk, tid);      // ends on a barrier
    ed_build_pairs(ptab, tab, tid);                // as well
    for (int it = 0; it < inner; it++) {
        uint32_t start = kEdNoStart;
        bool work = false;
        if (mine) {
            if (i == pk.first_sub) start = pk.bit0;
            else if (it == 0) start = pk.org + i * pk.sub_bits;       // a guess: the subsequence's own first bit
            else if (tid == 0) start = s_used[0];
            else start = s_end[tid - 1];
            work = s_used[tid] != start;
        }
        const unsigned long long mask = __ballot(work);
        if (lane == 0) s_wt[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t off = 0, n_work = 0;
        for (int w = 0; w < kEdThreads / 64; w++) { off += w < wave ? s_wt[w] : 0u; n_work += s_wt[w]; }
        if (n_work == 0) break;
        if (work) {
            const uint32_t slot = off + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            s_list[slot] = (uint32_t)tid;
            s_start[slot] = start;
        }
        __syncthreads();
        if ((uint32_t)tid < n_work) {
            const uint32_t l = s_list[tid], limit = ed_limit(pk, i0 + l);
            uint32_t count = 0;
            EdBitBuf r;
            r.start(lw, base, s_start[tid]);
            while (r.pos < limit) {
                uint32_t zeros, nb;
                int value;
                ed_run<true>(r, ptab, tab, cval, clen, zeros, nb, value);
                count += zeros + (nb ? 0x10001u : 0u);
            }
            s_used[l] = s_start[tid]; s_end[l] = r.pos; s_cnt[l] = count;
        }
        __syncthreads();
    }
    if (mine && (uint32_t)tid >= halo) {      // the halo's lanes belong to the workgroup in front
        const size_t at = (size_t)pk.sub_first + i;
        b.end[at] = s_end[tid];
        b.used[at] = s_used[tid];
        b.cnt[at] = s_cnt[tid];
    }
}

// The seams: one THREAD per workgroup of k_entd_sync.  That workgroup's first lane read from its guess; its true start is the end of the
// lane in front of it.  The thread reads the lane again from there and goes on into the lanes behind it until a read ends where the
// recorded one did (from there on nothing changes) -- one to three lanes, as a rule.  No table and no staged workgroup: the lane's few
// words go into the thread's own slice of LDS, the packet's 16 codes into registers, and codes are matched one by one (what a whole
// workgroup spent on staging and a table for ONE lane's read was half the time of the full pass: 160 us against 330 per twenty 4K
// packets; with the codes in LDS instead of registers the 32 dependent reads per code made this kernel slower than that: 420 us).
// A seam whose repair runs into the next seam's lanes while that one is being repaired is found by k_entd_verify (unsettled -> host parser).
constexpr int kEdFixThreads = 64;
constexpr uint32_t kEdFixWords = kEdMaxSubBits / 32 + 7;      // a lane's bits from its start (< 45 bits behind its first) to 45 + 64 behind its limit; odd: no bank conflicts
static_assert(kEdFixWords % 2 == 1, "per-thread LDS slices on an odd pitch");
// the tree code at the low end of w, from the packet's 16 codes in REGISTERS: c[s] = code | mask << 16 (a symbol without a code: 0xffff | 0,
// which nothing matches).  Exactly one code matches any bit pattern (a tree of >= 2 symbols is full).  Returns length | symbol << 4.
__device__ __forceinline__ uint32_t ed_code_regs(uint32_t w, const uint32_t (&c)[16])
{
    uint32_t e = 0x10000u;        // mask 1, symbol 0: ed_long_code's answer when nothing matches
#pragma unroll
    for (uint32_t s = 0; s < 16; s++)
        if ((w & (c[s] >> 16)) == (c[s] & 0xffffu)) e = (c[s] & 0xffff0000u) | s;
    return (uint32_t)__builtin_popcount(e >> 16) | ((e & 15u) << 4);
}
__global__ void __launch_bounds__(kEdFixThreads) k_entd_fix(EdBufs b, uint32_t n_groups)
{
    __shared__ uint32_t lw[kEdFixThreads][kEdFixWords];
    const int tid = (int)threadIdx.x;
    const uint32_t g = blockIdx.x * kEdFixThreads + (uint32_t)tid;
    if (g >= n_groups) return;
    const uint2 grp = b.groups[g];
    if (grp.y == 0) return;                        // a packet's first lane starts at its first run: true
    const EdPacket &pk = b.packets[grp.x];
    uint32_t i = grp.y * kEdOwn;
    if (i <= pk.first_sub) return;                 // the run streams begin in or behind this lane: it starts at bit0 (or has nothing to read)
    size_t at = (size_t)pk.sub_first + i;
    uint32_t start = __atomic_load_n(b.end + at - 1, __ATOMIC_RELAXED);
    if (b.used[at] == start) return;               // the guess was right
    uint32_t c[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const uint32_t l = pk.code_len[s];
        c[s] = l ? (uint32_t)pk.code_val[s] | (((1u << l) - 1u) << 16) : 0xffffu;
    }
    const uint32_t *src = (const uint32_t *)(b.bytes + pk.byte_off);
    const uint32_t have = (pk.total_bits + 31u) / 32u + 3u;
    // the repair stops at the end of the seam's OWN group of lanes: running on into the next group's lanes would have two threads of
    // different workgroups rewrite one lane's (used, end, cnt) triple at the same time; a wrong phase that lives longer than a group leaves the
    // next seam unsettled, k_entd_verify sees it and the packet goes to the host parser
    const uint32_t stop = min(pk.n_sub, (grp.y + 1u) * kEdOwn);
    for (; i < stop; i++, at++) {
        if (b.used[at] == start) break;
        const uint32_t limit = ed_limit(pk, i), w0 = start >> 5;
#pragma unroll
        for (uint32_t k = 0; k < kEdFixWords; k++) lw[tid][k] = w0 + k < have ? src[w0 + k] : 0u;
        uint32_t count = 0;
        EdReader r{lw[tid], w0 * 32u, start};
        while (r.pos < limit) {
            // one run without the table (ed_run's fields, code by code)
            uint32_t e = ed_code_regs(r.window(), c);
            r.pos += e & 15u;
            const uint32_t zeros = e >> 4;
            e = ed_code_regs(r.window(), c);
            const uint32_t nb = e >> 4;
            r.pos += (e & 15u) + nb;
            count += zeros + (nb ? 0x10001u : 0u);
        }
        const uint32_t old_end = b.end[at];
        b.used[at] = start;
        __atomic_store_n(b.end + at, r.pos, __ATOMIC_RELAXED);
        b.cnt[at] = count;
        if (r.pos == old_end) break;               // met the recorded read: the lanes behind are as they were
        start = r.pos;
    }
}

// one workgroup per entry of b.groups, nothing is read: a lane whose recorded start is not the end of the lane in front of it marks the
// packet (then end[0] was not followed through: the host parser reads it); the coefficients and values the workgroup's lanes cover,
// summed for k_entd_prefix.
__global__ void __launch_bounds__(kEdThreads) k_entd_verify(EdBufs b)
{
    __shared__ uint32_t s_end[kEdThreads];
    __shared__ unsigned long long scratch[kEdThreads];
    const uint2 grp = b.groups[blockIdx.x];
    const EdPacket &pk = b.packets[grp.x];
    const int tid = (int)threadIdx.x;
    const uint32_t i = grp.y * kEdOwn + (uint32_t)tid;
    const bool mine = tid < kEdOwn && i < pk.n_sub && i >= pk.first_sub;
    const size_t at = (size_t)pk.sub_first + i;
    uint32_t used = kEdNoStart, end = 0, count = 0;
    if (mine) { used = b.used[at]; end = b.end[at]; count = b.cnt[at]; }
    s_end[tid] = end;
    __syncthreads();
    if (mine) {
        const uint32_t start = i == pk.first_sub ? pk.bit0 : tid == 0 ? b.end[at - 1] : s_end[tid - 1];
        if (used != start) atomicOr(b.status + grp.x, kEdUnsettled);
    }
    unsigned long long sum = 0;
    (void)ed_block_exclusive(mine ? ed_split(count) : 0ull, scratch, tid, &sum);
    if (tid == 0) b.wgsum[b.group0 + blockIdx.x] = sum;
}

// ------------------------------------------------------------------ a p-frame's block headers (src/dec.rs:351-372), read on the device
// [has_mvec:1][has_coeff:1]([mx:7s][my:7s] if has_mvec) per macroblock, back to back from bit 152 on: a serial chain -- where header b + 1
// starts depends on header b's first bit -- and, until round 5, the last thing the decoders read on the HOST (0.1 ms per 4K packet and
// 3 bytes per macroblock over PCIe).  But every header is 2 or 16 bits: in units of 2 bits a header advances by 1 or by 8, so a reader that
// enters a chunk of 32 units (64 bits) does so at one of 8 offsets, and what it does from there -- how many headers it passes, how many of
// them coded, at which offset it enters the next chunk -- depends on the chunk's bits alone.  That is a map {0..7} -> {0..7} x counts per
// chunk; maps compose (associatively), so the true reader's way through all chunks is a scan over maps and not a guess (unlike the run
// streams, nothing here has to settle):
//   k_hdr_map   a workgroup takes 32 chunks: thread (c, e) walks chunk c from entry e; the 32 chunk maps are composed into the
//               workgroup's map (2 048 bits per map);
//   k_hdr_scan  one workgroup per packet: its workgroup maps (32 B each) in LDS, ONE thread follows entry 0 through them -- 380 dependent
//               LDS reads for a 4K frame -- and leaves every workgroup its true entry and the macroblock / coded-macroblock counts
//               before it; the workgroups behind the last header (the run streams' bits) are marked;
//   k_hdr_emit  the workgroups again: chunk maps recomputed, the true entry followed through the 32 chunks, then a thread per chunk walks
//               its headers and writes motion vectors and has_coeff of the macroblocks it passes.  The thread that passes the frame's
//               last macroblock knows where the run streams start and how many macroblocks are coded: it completes the packet
//               descriptor (bit0, total_coefs, list_cap, first_sub) for the k_entd_* kernels behind it.
// A payload that ends inside its headers, or has no bit left behind them although macroblocks are coded, is marked irregular: the host
// parser reads it (and decides what it is).
constexpr uint32_t kHdrBit0 = 152;               // 16 table bytes + 3 q-table indices (src/dec.rs:236-246)
constexpr uint32_t kHdrChunkUnits = 32;          // 64 bits
constexpr uint32_t kHdrChunksPerWg = kEdThreads / 8;
constexpr uint32_t kHdrWgBits = kHdrChunksPerWg * kHdrChunkUnits * 2;     // 2 048

// 96 payload bits from bit position p on (zeros behind the payload's words): a chunk's 64 and what its last header hangs over
struct HdrBits { uint32_t w0, w1, w2; };
__device__ __forceinline__ HdrBits hdr_load(const uint8_t *bytes, const EdPacket &pk, unsigned long long p)
{
    const uint32_t *src = (const uint32_t *)(bytes + pk.byte_off);
    const uint32_t have = (pk.total_bits + 31u) / 32u + 3u, k = (uint32_t)(p >> 5), sh = (uint32_t)p & 31u;
    uint32_t d[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) d[j] = k + j < have ? src[k + j] : 0u;
    return HdrBits{__builtin_amdgcn_alignbit(d[1], d[0], sh), __builtin_amdgcn_alignbit(d[2], d[1], sh), __builtin_amdgcn_alignbit(d[3], d[2], sh)};
}
// one chunk from entry e: exit offset | headers passed << 3 | coded among them << 9
__device__ __forceinline__ uint32_t hdr_walk(const HdrBits &x, uint32_t e)
{
    const unsigned long long v = (unsigned long long)x.w0 | ((unsigned long long)x.w1 << 32);
    uint32_t u = e, nb = 0, nc = 0;
    while (u < kHdrChunkUnits) {
        const uint32_t f = (uint32_t)(v >> (2u * u)) & 3u;      // bit 0: has_mvec, bit 1: has_coeff
        nb++;
        nc += f >> 1;
        u += (f & 1u) ? 8u : 1u;
    }
    return (u - kHdrChunkUnits) | (nb << 3) | (nc << 9);
}
// the chunk maps of a workgroup into LDS (cm[chunk][entry]) and, composed, its own map (thread 0..7: entry e) -- wm[e] = exit | headers << 3 |
// coded << 14; s_entry / s_nb / s_nc (optional): the way of ONE reader, entering at `e0`, through the chunks
__device__ __forceinline__ void hdr_wg_maps(uint16_t (*cm)[8], const uint8_t *bytes, const EdPacket &pk, uint32_t wg, uint32_t n_chunks, int tid)
{
    const uint32_t c = (uint32_t)tid >> 3, e = (uint32_t)tid & 7u, chunk = wg * kHdrChunksPerWg + c;
    uint32_t m = e;                                  // a chunk behind the payload passes a reader on as it came
    if (chunk < n_chunks) m = hdr_walk(hdr_load(bytes, pk, (unsigned long long)kHdrBit0 + (unsigned long long)chunk * 64ull), e);
    cm[c][e] = (uint16_t)m;
    __syncthreads();
}
__device__ __forceinline__ uint32_t hdr_chunks(const EdPacket &pk)
{
    const unsigned long long bits = pk.total_bits > kHdrBit0 ? (unsigned long long)(pk.total_bits - kHdrBit0) : 0ull;
    return (uint32_t)((min(bits, (unsigned long long)pk.total_blocks * 16ull) + 63ull) / 64ull);
}

// grid (header workgroups of the longest packet, packets of the window)
__global__ void __launch_bounds__(kEdThreads) k_hdr_map(EdBufs b)
{
    __shared__ uint16_t cm[kHdrChunksPerWg][8];
    const EdPacket &pk = b.packets[b.packet0 + blockIdx.y];
    if (blockIdx.x >= pk.hdr_wgs) return;
    const int tid = (int)threadIdx.x;
    hdr_wg_maps(cm, b.bytes, pk, blockIdx.x, hdr_chunks(pk), tid);
    if (tid < 8) {
        uint32_t x = (uint32_t)tid, nb = 0, nc = 0;
        for (uint32_t c = 0; c < kHdrChunksPerWg; c++) {
            const uint32_t m = cm[c][x];
            x = m & 7u; nb += (m >> 3) & 63u; nc += m >> 9;
        }
        b.hdr_maps[((size_t)pk.hdr_first + blockIdx.x) * 8u + (uint32_t)tid] = x | (nb << 3) | (nc << 14);
    }
}

// one workgroup per packet of the window
#ifndef PFV_HDR_SCAN_TILE
#define PFV_HDR_SCAN_TILE 1024                   // tests build a variant with a tile of 2 so that small frames take several tiles
#endif
constexpr uint32_t kHdrScanTile = PFV_HDR_SCAN_TILE;   // workgroup maps in LDS at a time (32 KiB: 2 Mbit of headers)
__global__ void __launch_bounds__(kEdThreads) k_hdr_scan(EdBufs b)
{
    __shared__ uint32_t wm[kHdrScanTile][8];
    __shared__ uint32_t carry[3];
    const EdPacket &pk = b.packets[b.packet0 + blockIdx.x];
    if (pk.hdr_wgs == 0) return;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { carry[0] = 0; carry[1] = 0; carry[2] = 0; }
    for (uint32_t t0 = 0; t0 < pk.hdr_wgs; t0 += kHdrScanTile) {
        const uint32_t n = min(kHdrScanTile, pk.hdr_wgs - t0);
        __syncthreads();
        for (uint32_t k = (uint32_t)tid; k < n * 8u; k += kEdThreads) wm[k >> 3][k & 7u] = b.hdr_maps[((size_t)pk.hdr_first + t0) * 8u + k];
        __syncthreads();
        if (tid == 0) {
            uint32_t e = carry[0], nb = carry[1], nc = carry[2];
            for (uint32_t w = 0; w < n; w++) {
                b.hdr_start[(size_t)pk.hdr_first + t0 + w] = make_uint4(e, nb, nc, 0u);
                const uint32_t m = wm[w][e];
                e = m & 7u; nb += (m >> 3) & 0x7ffu; nc += m >> 14;
            }
            carry[0] = e; carry[1] = nb; carry[2] = nc;
        }
    }
    __syncthreads();
    // fewer headers than macroblocks in all the bits there are: the payload ends inside its headers (the host parser reports it)
    if (tid == 0 && carry[1] < pk.total_blocks) atomicOr(b.status + b.packet0 + blockIdx.x, kEdIrregular);
}

// grid as k_hdr_map
__global__ void __launch_bounds__(kEdThreads) k_hdr_emit(EdBufs b)
{
    __shared__ uint16_t cm[kHdrChunksPerWg][8];
    __shared__ uint32_t c_entry[kHdrChunksPerWg], c_nb[kHdrChunksPerWg], c_nc[kHdrChunksPerWg];
    EdPacket &pk = b.packets[b.packet0 + blockIdx.y];
    if (blockIdx.x >= pk.hdr_wgs) return;
    const uint4 st = b.hdr_start[(size_t)pk.hdr_first + blockIdx.x];
    const uint32_t tb = pk.total_blocks;
    if (st.y >= tb) return;                          // behind the last header: these bits are run streams
    const int tid = (int)threadIdx.x;
    const uint32_t n_chunks = hdr_chunks(pk);
    hdr_wg_maps(cm, b.bytes, pk, blockIdx.x, n_chunks, tid);
    if (tid == 0) {
        uint32_t e = st.x, nb = st.y, nc = st.z;
        for (uint32_t c = 0; c < kHdrChunksPerWg; c++) {
            c_entry[c] = e; c_nb[c] = nb; c_nc[c] = nc;
            const uint32_t m = cm[c][e];
            e = m & 7u; nb += (m >> 3) & 63u; nc += m >> 9;
        }
    }
    __syncthreads();
    if ((uint32_t)tid >= kHdrChunksPerWg) return;
    const uint32_t chunk = blockIdx.x * kHdrChunksPerWg + (uint32_t)tid;
    uint32_t blk = c_nb[tid], nc = c_nc[tid];
    if (chunk >= n_chunks || blk >= tb) return;
    const unsigned long long p0 = (unsigned long long)kHdrBit0 + (unsigned long long)chunk * 64ull;
    const HdrBits x = hdr_load(b.bytes, pk, p0);
    int8_t *mv = b.mv + (pk.frame_off * tb) * 2u;
    uint8_t *has = b.has + pk.frame_off * tb;
    uint32_t u = c_entry[tid];
    while (u < kHdrChunkUnits && blk < tb) {
        // the header's 16 bits from the chunk's 96
        const uint32_t sh = 2u * u;
        const uint32_t lo = sh < 32u ? __builtin_amdgcn_alignbit(x.w1, x.w0, sh) : __builtin_amdgcn_alignbit(x.w2, x.w1, sh - 32u);
        const uint32_t w = sh == 32u ? x.w1 : lo;        // alignbit takes the shift modulo 32
        const uint32_t f = w & 3u;
        int mx = 0, my = 0;
        if (f & 1u) {
            mx = (int)(w << 23) >> 25;                   // bits 2..8, two's complement (src/dec.rs:366-369)
            my = (int)(w << 16) >> 25;                   // bits 9..15
        }
        mv[2u * blk] = (int8_t)mx; mv[2u * blk + 1u] = (int8_t)my;
        has[blk] = (uint8_t)(f >> 1);
        nc += f >> 1;
        blk++;
        u += (f & 1u) ? 8u : 1u;
        if (blk == tb) {
            // the frame's last header: the run streams start behind it
            const unsigned long long bit0 = p0 + 2ull * u;
            const uint32_t bits = pk.total_bits;
            if (bit0 > bits || (nc != 0 && bit0 >= bits)) {
                atomicOr(b.status + b.packet0 + blockIdx.y, kEdIrregular);     // ends inside its headers / nothing behind them: the host parser's case
                pk.total_coefs = 0;                                             // (nothing of it is read here)
            } else {
                pk.bit0 = (uint32_t)bit0;
                pk.total_coefs = nc * 256u;
                pk.first_sub = ((uint32_t)bit0 - pk.org) / pk.sub_bits;
                pk.list_cap = (uint32_t)min((unsigned long long)nc * 256ull, (unsigned long long)(bits - (uint32_t)bit0) / 3ull + 1ull);
            }
        }
    }
}

// one workgroup per packet: a p-frame's list of coded macroblocks (coded[k] = the k-th macroblock with has_coeff set, src/dec.rs:378-380)
// from its has_coeff bytes -- 4 bytes per macroblock that need not cross PCIe.  Rounds of 16 384 macroblocks: a thread takes 64 of them
// (four independent 16-byte loads, kept in registers), counts the non-zero bytes, and writes its macroblocks behind those of the threads
// and rounds before it.  (A first version read 64 bytes per wavefront and step with a ballot in between, a second read every byte again
// behind stores the compiler could not move the loads across: 99 and 56 us of waiting for one load after the other.)
__global__ void __launch_bounds__(kEdThreads) k_entd_coded(EdBufs b)
{
    __shared__ unsigned long long scratch[kEdThreads];
    const EdPacket &pk = b.packets[b.packet0 + blockIdx.x];
    if (!pk.pframe || pk.n_sub == 0 || pk.total_coefs == 0) return;
    const int tid = (int)threadIdx.x;
    const uint32_t tb = pk.total_blocks;
    const uint8_t *has = b.has + pk.frame_off * tb;
    uint32_t *coded = b.coded + pk.frame_off * tb;
    uint32_t carry = 0;
    for (uint32_t r0 = 0; r0 < tb; r0 += kEdThreads * 64u) {
        const uint32_t lo = min(tb, r0 + (uint32_t)tid * 64u), hi = min(tb, lo + 64u);
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const uint32_t m = lo + 16u * (uint32_t)q;
            if (m + 16u <= hi) {
                __builtin_memcpy(&v, has + m, 16);      // any alignment: a frame's row starts wherever total_blocks puts it
            } else if (m < hi) {
                uint8_t t[16] = {0};
                for (uint32_t k = m; k < hi; k++) t[k - m] = has[k];
                __builtin_memcpy(&v, t, 16);
            }
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        uint32_t n = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) n += (uint32_t)__builtin_popcount((w[q] | ((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u);   // non-zero bytes
        unsigned long long total = 0;
        uint32_t at = carry + (uint32_t)ed_block_exclusive((unsigned long long)n, scratch, tid, &total);
        carry += (uint32_t)total;
#pragma unroll
        for (int q = 0; q < 16; q++)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((w[q] >> (8 * k)) & 0xffu) coded[at++] = lo + 4u * (uint32_t)q + (uint32_t)k;
    }
}

// one workgroup per packet: wgsum[g] = coefficients covered by (and values of) the packet's workgroups before g
__global__ void __launch_bounds__(kEdThreads) k_entd_prefix(EdBufs b)
{
    __shared__ unsigned long long scratch[kEdThreads];
    const EdPacket &pk = b.packets[b.packet0 + blockIdx.x];
    const int tid = (int)threadIdx.x;
    if (pk.n_sub == 0 || (b.status[b.packet0 + blockIdx.x] & kEdUnsettled)) return;
    const uint32_t n = (pk.n_sub + kEdOwn - 1) / kEdOwn;
    unsigned long long *w = b.wgsum + pk.grp_first;
    unsigned long long carry = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += kEdThreads) {
        const uint32_t g = g0 + (uint32_t)tid;
        const unsigned long long v = g < n ? w[g] : 0ull;
        unsigned long long sum = 0;
        const unsigned long long ex = ed_block_exclusive(v, scratch, tid, &sum);
        if (g < n) w[g] = carry + ex;
        carry += sum;
    }
}

// What a workgroup of k_entd_emit stages in LDS: its entries (the workgroup's share of the packet's list is contiguous), the counts of the
// macroblocks that START inside it (contiguous in coefficient space) and, p-frames, the slice of the coded-macroblock list those need.  What
// does not fit -- content far denser than any real frame -- goes to memory directly.
#ifndef PFV_ED_OUT_CAP
#define PFV_ED_OUT_CAP 4096             // tests build a variant with tiny stages so that small frames take the straight-to-memory paths
#endif
#ifndef PFV_ED_MB_CAP
#define PFV_ED_MB_CAP 1024
#endif
constexpr uint32_t kEdOutCap = PFV_ED_OUT_CAP;    // entries
constexpr uint32_t kEdMbCap = PFV_ED_MB_CAP;      // macroblock starts

// counts[m] = n for the macroblocks (prev, mb]: the coded macroblock `mb` and the skipped ones in front of it own nothing before entry n
__device__ __forceinline__ void ed_fill_counts(uint32_t *counts, uint32_t prev, uint32_t mb, uint32_t n)
{
    for (uint32_t m = prev + 1u; m <= mb; m++) counts[m] = n;       // prev == 0xffffffff: from macroblock 0
}

// workgroups as k_entd_sync: the values of subsequence i into the packet's coefficient list, the frame's counts beside them
__global__ void __launch_bounds__(kEdThreads) k_entd_emit(EdBufs b)
{
    __shared__ __attribute__((aligned(16))) uint8_t tab[4096];
    __shared__ uint16_t cval[16];
    __shared__ uint8_t clen[16];
    __shared__ uint32_t lw[kEdStageWords];
    __shared__ unsigned long long scratch[kEdThreads];
    __shared__ uint32_t s_out[kEdOutCap];
    __shared__ uint32_t s_start[kEdMbCap];
    __shared__ uint32_t s_mb[kEdMbCap + 1];
    const uint2 grp = b.groups[blockIdx.x];
    const EdPacket &pk = b.packets[grp.x];
    if (b.status[grp.x] & kEdUnsettled) return;           // set by an earlier launch (the whole workgroup leaves)
    const int tid = (int)threadIdx.x;
    const uint32_t i = grp.y * kEdOwn + (uint32_t)tid;
    const uint32_t base = ed_stage(lw, b.bytes, pk, grp.y * kEdOwn, tid);
    ed_build_table(tab, cval, clen, pk, tid);
    const bool mine = tid < kEdOwn && i < pk.n_sub && i >= pk.first_sub;
    const size_t at = (size_t)pk.sub_first + i;
    // what the packet's workgroups before this one cover, what this one's lanes do, and this lane's place among them: coefficients (low
    // half) and values (high half)
    const unsigned long long wg0 = b.wgsum[b.group0 + blockIdx.x];
    unsigned long long wg_sum = 0;
    const unsigned long long before = wg0 + ed_block_exclusive(mine ? ed_split(b.cnt[at]) : 0ull, scratch, tid, &wg_sum);
    const uint32_t total = pk.total_coefs, cap = pk.list_cap, n_k = total >> 8;
    const uint32_t Vwg = (uint32_t)wg0, Owg = (uint32_t)(wg0 >> 32), n_wg = (uint32_t)(wg_sum >> 32);
    // the macroblocks whose first coefficient lies in the workgroup: [ka, kb) in coefficient space (the k-th CODED macroblock of a p-frame)
    const uint32_t ka = (uint32_t)min((unsigned long long)n_k, ((wg0 & 0xffffffffull) + 255ull) >> 8);
    const uint32_t kb = (uint32_t)min((unsigned long long)n_k, ((wg0 & 0xffffffffull) + (wg_sum & 0xffffffffull) + 255ull) >> 8);
    const uint32_t *coded = b.coded + pk.frame_off * pk.total_blocks;
    uint32_t *counts = b.counts + pk.frame_off * (pk.total_blocks + 1u);
    uint32_t *list = b.lists[pk.frame_off];
    if (pk.pframe) {   // s_mb[j] = coded[ka - 1 + j]: the macroblocks that start here and the one the workgroup's first run may still be inside
        const uint32_t n_stage = min(kb - ka + 1u, kEdMbCap + 1u);
        for (uint32_t j = (uint32_t)tid; j < n_stage; j += kEdThreads) s_mb[j] = ka + j >= 1u ? coded[ka + j - 1u] : 0xffffffffu;
        __syncthreads();
    }
    uint32_t V = (uint32_t)before;             // the coefficient index the lane's first run starts at
    uint32_t O = (uint32_t)(before >> 32);     // the list index of its first value
    const bool active = mine && (before & 0xffffffffull) < total;     // behind the last coefficient nothing is read (src/dec.rs:261, :382)
    bool odd = false;
    if (active) {
        const uint32_t start = i == pk.first_sub ? pk.bit0 : b.end[at - 1];
        const uint32_t limit = ed_limit(pk, i);
        EdReader r{lw, base, start};
        if (pk.pframe) {
            // one run stream per coded macroblock, closed exactly on its 256th coefficient: a macroblock's first run starts at V % 256 == 0
            auto mb_of = [&](uint32_t k) -> uint32_t { const uint32_t j = k + 1u - ka; return j <= kEdMbCap ? s_mb[j] : coded[k]; };
            uint32_t mb = (V & 255u) ? mb_of(V >> 8) : 0u;     // the macroblock V lies in
            while (r.pos < limit && V < total) {
                uint32_t zeros, nb;
                int value;
                if (!(V & 255u)) {                                     // a macroblock starts: O values lie before it
                    const uint32_t k = V >> 8, j = k - ka;
                    mb = mb_of(k);
                    if (j < kEdMbCap) s_start[j] = O;
                    else ed_fill_counts(counts, k ? mb_of(k - 1u) : 0xffffffffu, mb, O);
                }
                ed_run(r, tab, cval, clen, zeros, nb, value);
                if (r.pos > pk.total_bits) { odd = true; break; }     // the run's fields run past the payload
                const uint32_t local = (V & 255u) + zeros;
                V += zeros;
                if (local >= 256u) {                                    // the run closes the macroblock: exactly, and without a value
                    if (local != 256u || nb) { odd = true; break; }
                    continue;
                }
                if (nb) {
                    const uint32_t e = coef_entry(mb, local, (int16_t)value), ol = O - Owg;
                    if (ol < kEdOutCap) s_out[ol] = e;
                    else if (O < cap) list[O] = e;
                    O++;
                    V++;
                }
            }
            if (V >= total && !odd) ed_fill_counts(counts, mb_of(n_k - 1u), pk.total_blocks, O);   // behind the last coded macroblock: all values
        } else {
            // ONE run stream over all macroblocks: the run that reaches or crosses the boundary 256 k -- V <= 256 k < V + zeros + (a value ? 1 : 0)
            // -- knows how many values lie before macroblock k
            while (r.pos < limit && V < total) {
                uint32_t zeros, nb;
                int value;
                ed_run(r, tab, cval, clen, zeros, nb, value);
                if (r.pos > pk.total_bits) { odd = true; break; }
                const uint32_t k = (V + 255u) >> 8, after = V + zeros + (nb ? 1u : 0u);
                if ((k << 8) < after && k < n_k) {                      // zeros <= 15: at most one boundary per run
                    if (k - ka < kEdMbCap) s_start[k - ka] = O;
                    else counts[k] = O;
                }
                V += zeros;
                if (V >= total) {                                       // the closing run of the frame
                    if (nb) odd = true;
                    break;
                }
                if (nb) {
                    const uint32_t e = coef_entry(V >> 8, V & 255u, (int16_t)value), ol = O - Owg;
                    if (ol < kEdOutCap) s_out[ol] = e;
                    else if (O < cap) list[O] = e;
                    O++;
                    V++;
                }
            }
            if (V >= total && !odd) counts[n_k] = O;                   // the lane whose run reached the last coefficient
        }
        if (!odd && i + 1 == pk.n_sub && V < total) odd = true;      // the payload ends before the last coefficient
        if (!odd && O > cap) odd = true;                              // cannot happen for a payload the host parser accepts (entd_list_cap); never written past
    }
    if (odd) atomicOr(b.status + grp.x, kEdIrregular);
    __syncthreads();
    // the staged entries and counts leave as whole lines
    const uint32_t n_out = min(n_wg, kEdOutCap);
    for (uint32_t j = (uint32_t)tid; j < n_out; j += kEdThreads)
        if (Owg + j < cap) list[Owg + j] = s_out[j];
    const uint32_t n_mb = min(kb - ka, kEdMbCap);
    if (pk.pframe) {
        for (uint32_t j = (uint32_t)tid; j < n_mb; j += kEdThreads) ed_fill_counts(counts, s_mb[j], s_mb[j + 1u], s_start[j]);
    } else {
        for (uint32_t j = (uint32_t)tid; j < n_mb; j += kEdThreads) counts[ka + j] = s_start[j];
    }
}

}  // namespace pfv
