// pfv_entdec_kernels.hip -- the decoder's entropy stage on the device (gfx950): packet payloads -> dense coefficient arrays.
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
//                 LDS until nothing changes) and are launches between workgroups.  A last launch only verifies that no lane has
//                 anything left to do: then end[0] is true (the first lane starts at the true first run) and every end[i] follows
//                 from a true start.  A packet that has not settled (periodic content can keep a wrong phase for ever) is left to
//                 the host parser.
//   k_entd_prefix exclusive prefix over the coefficients each subsequence covers: the coefficient index its first run starts at.
//   k_entd_emit   every lane reads its subsequence once more, from its true start and coefficient index, and stores the values
//                 (zeros are what the buffer was cleared to).  It also decides whether the host parser would have accepted the
//                 payload and produced the same array: anything it is not sure of -- a field that runs past the payload, a value
//                 behind the last coefficient, a macroblock whose runs do not end exactly on its 256th coefficient -- marks the
//                 packet, and a marked packet is parsed by the host code instead (pfv_host.hip: read_runs), which alone
//                 decides about errors.  The device path therefore never has to reproduce an error case.
//
// Code pairs are looked up in a 12-bit table (LDS, built by the workgroup from the packet's 16 codes); longer pairs and codes
// go through the 16 codes one by one.  Included by pfv_capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfv {

constexpr int kEdThreads = 256;
#ifndef PFV_ED_SUB_BITS
#define PFV_ED_SUB_BITS 256
#endif
constexpr uint32_t kEdSubBits = PFV_ED_SUB_BITS;     // payload bits per lane (default; EdPacket::sub_bits is what the kernels use)
constexpr uint32_t kEdIrregular = 1u;               // k_entd_emit: the host parser decides about this packet
constexpr uint32_t kEdUnsettled = 2u;               // k_entd_sync: the subsequence starts had not settled
constexpr uint32_t kEdNoStart = 0xffffffffu;
constexpr int kEdInner = 24;                        // k_entd_sync: rounds inside a workgroup per launch (default)

// one packet of the batch (made by the host from the packet's first 19 bytes and, p-frames, its block headers)
struct EdPacket {
    unsigned long long byte_off;   // payload position in the device byte buffer (multiple of 4; >= 8 readable bytes behind the payload)
    uint32_t total_bits;           // payload size in bits
    uint32_t bit0;                 // first bit of the run streams (behind the table, the q indices and the block headers)
    uint32_t total_coefs;          // coefficients the run streams cover: macroblocks x 256 (i-frame), coded macroblocks x 256 (p-frame)
    uint32_t n_sub;                // subsequences = ceil((total_bits - bit0) / sub_bits); 0: nothing to read
    uint32_t sub_bits;             // payload bits per lane
    uint32_t sub_first;            // index of subsequence 0 in the per-subsequence arrays
    uint32_t pframe;               // 1: values go through the coded-macroblock list
    uint32_t total_blocks;         // macroblocks per frame
    unsigned long long frame_off;  // which frame of the has / coded-list / coefficient arrays this packet fills
    uint16_t code_val[16];         // the packet's tree codes, LSB first (src/huffman.rs:204-217)
    uint8_t code_len[16];          // 0: the symbol has no code
};

struct EdBufs {
    const uint8_t *bytes;          // payloads
    const EdPacket *packets;
    const uint2 *groups;           // workgroups of k_entd_sync / k_entd_emit: (packet, which kEdThreads subsequences of it)
    uint32_t *end, *used, *cnt, *vstart;   // per subsequence
    const uint32_t *coded;         // [frame][total_blocks]: the k-th coded macroblock of the frame (p-frames; from the host's pass over the block headers)
    int16_t *coef;                 // [frame][total_blocks][256], cleared
    uint32_t *status;              // per packet: kEd* bits
    uint32_t packet0;              // k_entd_prefix: the launch's first packet (one workgroup per packet)
};

// (used bits | num_zeroes << 4 | coeff_size << 8) of the code pair at the low end of v, 0 when the pair is longer than 12 bits
__device__ __forceinline__ void ed_build_pairs(uint16_t *pair, uint16_t *cval, uint8_t *clen, const EdPacket &pk, int tid)
{
    if (tid < 16) { cval[tid] = pk.code_val[tid]; clen[tid] = pk.code_len[tid]; }
    __syncthreads();
    for (uint32_t v = (uint32_t)tid; v < 4096u; v += kEdThreads) {
        uint32_t e = 0, la = 0, za = 0;
        for (uint32_t s = 0; s < 16; s++) {
            const uint32_t l = clen[s];
            if (l && l <= 11 && (v & ((1u << l) - 1u)) == cval[s]) { la = l; za = s; }
        }
        if (la) {
            const uint32_t r = v >> la;
            for (uint32_t s = 0; s < 16; s++) {
                const uint32_t l = clen[s];
                if (l && la + l <= 12 && (r & ((1u << l) - 1u)) == cval[s]) e = (la + l) | (za << 4) | (s << 8);
            }
        }
        pair[v] = (uint16_t)e;
    }
    __syncthreads();
}

struct EdReader {
    const uint32_t *words;
    uint64_t buf;
    uint32_t have, wi, pos;
    __device__ __forceinline__ void open(const uint8_t *payload, uint32_t at)
    {
        words = (const uint32_t *)payload;
        pos = at;
        wi = at >> 5;
        buf = (uint64_t)words[wi++] >> (at & 31u);
        have = 32u - (at & 31u);
    }
    __device__ __forceinline__ void refill()   // >= 33 valid bits behind this
    {
        if (have <= 32u) {
            buf |= (uint64_t)words[wi++] << have;
            have += 32u;
        }
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; have -= n; pos += n; }
};

// one tree code through the 16 codes (a tree of >= 2 symbols is full: exactly one code matches any bit pattern)
__device__ __forceinline__ uint32_t ed_code(EdReader &r, const uint16_t *cval, const uint8_t *clen)
{
    r.refill();
    const uint32_t w = (uint32_t)r.buf;
    uint32_t sym = 0, len = 1;
    for (uint32_t s = 0; s < 16; s++) {
        const uint32_t l = clen[s];
        if (l && (w & ((1u << l) - 1u)) == cval[s]) { sym = s; len = l; }
    }
    r.drop(len);
    return sym;
}

// one run: num_zeroes, coeff_size, the value bits (sign-extended); the reader ends up on the next run
__device__ __forceinline__ void ed_run(EdReader &r, const uint16_t *pair, const uint16_t *cval, const uint8_t *clen, uint32_t &zeros, uint32_t &nb, int &value)
{
    r.refill();
    const uint32_t e = pair[(uint32_t)r.buf & 4095u];
    if (e) {
        zeros = (e >> 4) & 15u;
        nb = e >> 8;
        r.drop(e & 15u);           // <= 12 of >= 33: the value's <= 15 bits are there
    } else {
        zeros = ed_code(r, cval, clen);
        nb = ed_code(r, cval, clen);
        r.refill();
    }
    const uint32_t raw = (uint32_t)r.buf & ((1u << nb) - 1u), sign = (1u << nb) >> 1;
    value = (int)((raw ^ sign) - sign);
    r.drop(nb);
}

__device__ __forceinline__ uint32_t ed_limit(const EdPacket &pk, uint32_t i)
{
    const unsigned long long lim = (unsigned long long)pk.bit0 + (unsigned long long)(i + 1u) * pk.sub_bits;
    return lim < pk.total_bits ? (uint32_t)lim : pk.total_bits;
}

// one workgroup per entry of b.groups.  Inside a launch the lanes of a workgroup pass their ends along through LDS and repeat until
// none of them has a new start (a lane whose read had not met the true one by its end changes its neighbour's start, and so on: a few
// short rounds instead of launches); between workgroups the ends travel through memory, launch to launch.  verify != 0: nothing is
// read, a lane that still has work marks the packet.
__global__ void __launch_bounds__(kEdThreads) k_entd_sync(EdBufs b, int first_round, int verify, int inner)
{
    __shared__ uint16_t pair[4096];
    __shared__ uint16_t cval[16];
    __shared__ uint8_t clen[16];
    __shared__ uint32_t s_end[kEdThreads];
    __shared__ int any_work;
    const uint2 grp = b.groups[blockIdx.x];
    const EdPacket &pk = b.packets[grp.x];
    const int tid = (int)threadIdx.x;
    const uint32_t i = grp.y * kEdThreads + (uint32_t)tid;
    const bool mine = i < pk.n_sub;
    const size_t at = (size_t)pk.sub_first + i;
    uint32_t used = kEdNoStart, end = 0, count = 0, before = 0;
    if (mine && !first_round) { used = b.used[at]; end = b.end[at]; count = b.cnt[at]; }
    if (mine && !first_round && tid == 0 && i > 0) before = b.end[at - 1];       // the workgroup in front: as the last launch left it
    const uint32_t limit = mine ? ed_limit(pk, i) : 0;
    bool built = false, dirty = false;
    for (int it = 0; it < inner; it++) {
        s_end[tid] = end;
        if (tid == 0) any_work = 0;
        __syncthreads();
        uint32_t start = kEdNoStart;
        bool work = false;
        if (mine) {
            if (i == 0) start = pk.bit0;
            else if (first_round && it == 0) start = pk.bit0 + i * pk.sub_bits;  // a guess: the subsequence's own first bit
            else if (tid == 0) start = first_round ? used : before;
            else start = s_end[tid - 1];
            work = used != start;
        }
        if (work) any_work = 1;
        __syncthreads();
        if (!any_work) break;
        if (verify) {
            if (work) atomicOr(b.status + grp.x, kEdUnsettled);
            return;
        }
        if (!built) {
            ed_build_pairs(pair, cval, clen, pk, tid);
            built = true;
        }
        if (work) {
            count = 0;
            EdReader r;
            r.open(b.bytes + pk.byte_off, start);
            while (r.pos < limit) {
                uint32_t zeros, nb;
                int value;
                ed_run(r, pair, cval, clen, zeros, nb, value);
                count += zeros + (nb ? 1u : 0u);
            }
            end = r.pos;
            used = start;
            dirty = true;
        }
        __syncthreads();
    }
    if (dirty) {
        b.end[at] = end;
        b.used[at] = used;
        b.cnt[at] = count;
    }
}

// workgroup scan helper: exclusive prefix of one value per lane (kEdThreads lanes), total in *sum
__device__ __forceinline__ uint32_t ed_block_exclusive(uint32_t v, uint32_t *scratch, int tid, uint32_t *sum)
{
    scratch[tid] = v;
    __syncthreads();
    for (int d = 1; d < kEdThreads; d <<= 1) {
        const uint32_t add = tid >= d ? scratch[tid - d] : 0u;
        __syncthreads();
        scratch[tid] += add;
        __syncthreads();
    }
    const uint32_t incl = scratch[tid];
    if (sum) *sum = scratch[kEdThreads - 1];
    __syncthreads();
    return incl - v;
}

// one workgroup per packet: vstart[i] = coefficients covered by the subsequences before i
__global__ void __launch_bounds__(kEdThreads) k_entd_prefix(EdBufs b)
{
    __shared__ uint32_t scratch[kEdThreads];
    const EdPacket &pk = b.packets[b.packet0 + blockIdx.x];
    const int tid = (int)threadIdx.x;
    if (pk.n_sub == 0) return;
    const uint32_t per = (pk.n_sub + kEdThreads - 1) / kEdThreads;
    const uint32_t lo = min((uint32_t)tid * per, pk.n_sub), hi = min(lo + per, pk.n_sub);
    const uint32_t *cnt = b.cnt + pk.sub_first;
    uint32_t *vs = b.vstart + pk.sub_first;
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += cnt[i];
    uint32_t run = ed_block_exclusive(mine, scratch, tid, nullptr);
    for (uint32_t i = lo; i < hi; i++) {
        vs[i] = run;
        run += cnt[i];
    }
}

// workgroups as k_entd_sync: the values of subsequence i into the coefficient array
__global__ void __launch_bounds__(kEdThreads) k_entd_emit(EdBufs b)
{
    __shared__ uint16_t pair[4096];
    __shared__ uint16_t cval[16];
    __shared__ uint8_t clen[16];
    const uint2 grp = b.groups[blockIdx.x];
    const EdPacket &pk = b.packets[grp.x];
    if (b.status[grp.x] & kEdUnsettled) return;           // set by an earlier launch
    const int tid = (int)threadIdx.x;
    const uint32_t i = grp.y * kEdThreads + (uint32_t)tid;
    ed_build_pairs(pair, cval, clen, pk, tid);
    if (i >= pk.n_sub) return;
    const size_t at = (size_t)pk.sub_first + i;
    const uint32_t start = i == 0 ? pk.bit0 : b.end[at - 1];
    const uint32_t limit = ed_limit(pk, i), total = pk.total_coefs;
    uint32_t V = b.vstart[at];
    const uint32_t *coded = b.coded + pk.frame_off * pk.total_blocks;
    int16_t *coef = b.coef + pk.frame_off * pk.total_blocks * 256u;
    bool odd = false;
    EdReader r;
    r.open(b.bytes + pk.byte_off, start);
    while (r.pos < limit && V < total) {
        uint32_t zeros, nb;
        int value;
        ed_run(r, pair, cval, clen, zeros, nb, value);
        if (r.pos > pk.total_bits) { odd = true; break; }            // the run's fields run past the payload
        if (pk.pframe) {
            const uint32_t local = (V & 255u) + zeros;
            if (local >= 256u) {                                       // the run closes the macroblock: exactly, and without a value
                if (local != 256u || nb) { odd = true; break; }
                V += zeros;
                continue;
            }
            V += zeros;
            if (nb) {
                coef[(size_t)coded[V >> 8] * 256u + (V & 255u)] = (int16_t)value;
                V++;
            }
        } else {
            V += zeros;
            if (V >= total) {                                          // the closing run of the frame
                if (nb) odd = true;
                break;
            }
            if (nb) coef[V++] = (int16_t)value;
        }
    }
    if (!odd && i + 1 == pk.n_sub && V < total) odd = true;          // the payload ends before the last coefficient
    if (odd) atomicOr(b.status + grp.x, kEdIrregular);
}

}  // namespace pfv
