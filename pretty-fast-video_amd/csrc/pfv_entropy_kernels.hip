// pfv_entropy_kernels.hip -- the encoder's entropy stage on the device (gfx950).
//
// The reference serialises a frame on one host thread (src/enc.rs:237-320 i-frames, :332-470 p-frames): rle_encode per
// macroblock (src/rle.rs:9-47), one 16-symbol histogram per frame (rle.rs:40-47), a Huffman tree from the normalised
// histogram (rle.rs:49-66, src/huffman.rs:71-119) and LSB-first bit packing.  At device rates that host stage is the
// whole cost of the encoder (a 1080p frame: ~30 us of kernels vs milliseconds of host entropy + 6 MB of PCIe), so
// the same byte stream is produced here, from the coefficient / header buffers the encode kernels left in HBM:
//
//   k_ent_scan     one lane per 8x8 subblock: non-zero bitmap, per-subblock symbol counts (16 x 8-bit) and the sum of
//                  coefficient sizes; block-reduced into the frame histogram of each stream
//   k_ent_codes    one workgroup per stream: histogram -> table bytes -> Huffman codes (the reference's construction,
//                  tie-breaks included) -> 256 pre-joined (num_zeroes, coeff_size) code pairs
//   k_ent_offsets  one workgroup per stream: bits per subblock = counts . code lengths + sizes, exclusive prefix
//                  sums for the block-header section (p-frames) and the symbol section; payload size
//   k_ent_init     zeroes exactly the words the payload will occupy and writes the 19 header bytes
//   k_ent_pack     one lane per subblock: walks the non-zero bitmap again and writes its bits at its offset (first and
//                  last word with atomicOr, interior words with plain stores); p-frame block headers likewise
//
// Run order inside a macroblock is the coefficient buffer's own (zigzag within a subblock, subblocks 0..3), runs cross
// subblock boundaries and end at the macroblock (enc.rs:246-255), so lane (mb, sb) needs only the bitmaps of the
// earlier subblocks of its macroblock.  Included by pfv_capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfv {

constexpr int kEntThreads = 256;
constexpr int kEntScanThreads = 1024;
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

struct EntFrame {
    int total_blocks;          // macroblocks per stream (Y then U then V)
    int n_streams;
    int pframe;                // block headers present, has_coef honoured
    uint32_t cap_bytes;        // payload capacity per stream (multiple of 4)
    uint8_t qidx[3];
    uint8_t pad;
};

struct EntBufs {
    const int16_t *coef;       // [S][total_blocks][256]
    const int8_t *mv;          // [S][total_blocks][2]   (p-frames)
    const uint8_t *has;        // [S][total_blocks]      (p-frames)
    uint64_t *mask;            // [S][total_blocks*4] non-zero bitmap per subblock (0 for uncoded macroblocks)
    uint4 *counts;             // [S][total_blocks*4] 16 x 8-bit symbol counts
    uint32_t *sumsize;         // [S][total_blocks*4] sum of coeff_size over the subblock's values
    uint32_t *sb_off;          // [S][total_blocks*4] bit offset of the subblock's symbols in the payload
    uint32_t *hdr_off;         // [S][total_blocks]   bit offset of the block header (p-frames)
    int32_t *hist;             // [S][16]
    EntCodes *codes;           // [S]
    uint32_t *sizes;           // [S] payload bytes or kEntErr*
    uint8_t *payload;          // [S][cap_bytes]
};

__device__ __forceinline__ uint32_t ent_shfl(uint32_t v, int src_lane)
{
    return (uint32_t)__shfl((int)v, src_lane);
}
__device__ __forceinline__ uint32_t ent_wave_sum(uint32_t v)
{
    for (int m = 1; m < 64; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}

// 8-bit counters for the 16 symbols in two 64-bit words
struct SymCount {
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ void add(unsigned bin, unsigned n)
    {
        const uint64_t inc = (uint64_t)n << (8u * (bin & 7u));
        if (bin < 8u) lo += inc;
        else hi += inc;
    }
};

// Position of the last non-zero coefficient before subblock `sb` in macroblock order (-1: none), from the quad's bitmaps.
__device__ __forceinline__ int ent_prev_last(uint64_t mine, int sb)
{
    const int lane = (int)(threadIdx.x & 63u), q0 = lane & ~3;
    int last = -1;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t lo = ent_shfl((uint32_t)mine, q0 + j), hi = ent_shfl((uint32_t)(mine >> 32), q0 + j);
        const uint64_t m = ((uint64_t)hi << 32) | lo;
        if (j < sb && m) last = 64 * j + 63 - __builtin_clzll(m);
    }
    return last;
}

// fillers (15, 0) needed before a run of `run` zeros can be coded in 4 bits, and what is left (rle.rs:18-21, 31-34)
__device__ __forceinline__ void ent_split_run(unsigned run, unsigned &fillers, unsigned &rest)
{
    fillers = run > 15u ? (run - 1u) / 15u : 0u;
    rest = run - 15u * fillers;
}

// ---------------------------------------------------------------------------------------------------- k_ent_scan
__global__ void __launch_bounds__(kEntThreads) k_ent_scan(EntFrame f, EntBufs b)
{
    __shared__ uint32_t blk[8];
    const int stream = (int)blockIdx.y, n_sb = f.total_blocks * 4;
    const int sbi = (int)(blockIdx.x * kEntThreads + threadIdx.x);
    const bool live = sbi < n_sb;
    const int mb = sbi >> 2, sb = sbi & 3;
    if (threadIdx.x < 8) blk[threadIdx.x] = 0;
    __syncthreads();

    const size_t sbase = (size_t)stream * n_sb;
    const int16_t *c = b.coef + ((size_t)stream * f.total_blocks + (live ? mb : 0)) * 256 + sb * 64;
    const bool coded = live && (!f.pframe || b.has[(size_t)stream * f.total_blocks + mb] != 0);
    uint64_t mask = 0;
    if (coded) {
        const uint4 *c4 = (const uint4 *)c;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 v = c4[k];
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t two = ((d[j] & 0xffffu) ? 1u : 0u) | ((d[j] >> 16) ? 2u : 0u);
                mask |= (uint64_t)two << (8 * k + 2 * j);
            }
        }
    }
    int last = ent_prev_last(mask, sb);   // every lane of the wavefront takes part in the exchange

    SymCount cnt;
    uint32_t sumsize = 0, oversize = 0;
    for (uint64_t mm = mask; mm;) {
        const int bit = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int i = 64 * sb + bit;
        unsigned fillers, rest;
        ent_split_run((unsigned)(i - last - 1), fillers, rest);
        last = i;
        const int v = c[bit];
        const unsigned mag = (unsigned)(v < 0 ? -v : v);
        const unsigned size = 33u - (unsigned)__builtin_clz(mag);   // bit length + 1 (rle.rs:23-24)
        oversize |= size > 15u;
        cnt.add(15u, fillers);
        cnt.add(0u, fillers);
        cnt.add(rest, 1u);
        cnt.add(size & 15u, 1u);
        sumsize += size;
    }
    if (coded && sb == 3 && last < 255) {   // the trailing run closes the macroblock (rle.rs:31-38)
        unsigned fillers, rest;
        ent_split_run((unsigned)(255 - last), fillers, rest);
        cnt.add(15u, fillers);
        cnt.add(0u, fillers + 1u);
        cnt.add(rest, 1u);
    }
    if (live) {
        b.mask[sbase + sbi] = mask;
        b.counts[sbase + sbi] = make_uint4((uint32_t)cnt.lo, (uint32_t)(cnt.lo >> 32), (uint32_t)cnt.hi, (uint32_t)(cnt.hi >> 32));
        b.sumsize[sbase + sbi] = sumsize;
    }

    // frame histogram: widen to 16-bit fields (two symbols per word; a workgroup adds at most 256 * 164 per field)
    const uint32_t w8[4] = {(uint32_t)cnt.lo, (uint32_t)(cnt.lo >> 32), (uint32_t)cnt.hi, (uint32_t)(cnt.hi >> 32)};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t a = (w8[j] & 0xffu) | ((w8[j] & 0xff00u) << 8);
        const uint32_t c2 = ((w8[j] >> 16) & 0xffu) | ((w8[j] >> 24) << 16);
        const uint32_t sa = ent_wave_sum(a), sc = ent_wave_sum(c2);
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&blk[2 * j], sa);
            atomicAdd(&blk[2 * j + 1], sc);
        }
    }
    const uint32_t any_over = ent_wave_sum(oversize);
    if ((threadIdx.x & 63u) == 0 && any_over) atomicOr(&b.codes[stream].oversize, 1u);
    __syncthreads();
    if (threadIdx.x < 16) {
        const uint32_t w = blk[threadIdx.x >> 1];
        const uint32_t n = (threadIdx.x & 1u) ? (w >> 16) : (w & 0xffffu);
        if (n) atomicAdd(&b.hist[stream * 16 + (int)threadIdx.x], (int32_t)n);
    }
}

// ---------------------------------------------------------------------------------------------------- k_ent_codes
// HuffmanTree::from_table (huffman.rs:71-119) + assign_codes (:204-217) on one lane; the sort is stable and the merged
// node goes in front of the first strictly smaller entry (:61-69), exactly as on the host (pfv_host.hip HuffmanTree).
__device__ inline void ent_build_codes(const int32_t *hist, uint8_t *table, uint32_t *val, uint8_t *len)
{
    int32_t mx = 0;
    for (int i = 0; i < 16; i++) mx = hist[i] > mx ? hist[i] : mx;
    for (int i = 0; i < 16; i++) {   // rle_create_huffman (rle.rs:49-66)
        uint32_t t = 0;
        if (hist[i] > 0) {
            t = (uint32_t)(((uint64_t)(uint32_t)hist[i] * 255u) / (uint32_t)mx);
            t = t < 1u ? 1u : t;
        }
        table[i] = (uint8_t)t;
        val[i] = 0;
        len[i] = 0;
    }
    uint32_t freq[16];
    int node[16], n = 0, n_nodes = 0;
    int left[31], right[31], sym[31];
    for (int ch = 0; ch < 16; ch++)
        if (table[ch]) {
            sym[n_nodes] = ch;
            left[n_nodes] = right[n_nodes] = -1;
            freq[n] = table[ch];
            node[n++] = n_nodes++;
        }
    for (int i = 1; i < n; i++) {   // stable, descending
        const uint32_t fq = freq[i];
        const int nd = node[i];
        int j = i - 1;
        for (; j >= 0 && freq[j] < fq; j--) { freq[j + 1] = freq[j]; node[j + 1] = node[j]; }
        freq[j + 1] = fq;
        node[j + 1] = nd;
    }
    while (n > 1) {
        const int a = node[n - 1], bnode = node[n - 2];
        const uint32_t fq = freq[n - 1] + freq[n - 2];
        n -= 2;
        sym[n_nodes] = -1;
        left[n_nodes] = a;
        right[n_nodes] = bnode;
        int pos = 0;
        while (pos < n && !(fq > freq[pos])) pos++;
        for (int j = n; j > pos; j--) { freq[j] = freq[j - 1]; node[j] = node[j - 1]; }
        freq[pos] = fq;
        node[pos] = n_nodes++;
        n++;
    }
    if (n == 0) return;
    int st_node[32];
    uint32_t st_val[32];
    uint8_t st_len[32];
    int sp = 0;
    st_node[0] = node[0]; st_val[0] = 0; st_len[0] = 0; sp = 1;
    while (sp) {
        sp--;
        const int nd = st_node[sp];
        const uint32_t v = st_val[sp];
        const uint8_t l = st_len[sp];
        if (sym[nd] >= 0) {
            val[sym[nd]] = v;
            len[sym[nd]] = l;
            continue;
        }
        st_node[sp] = left[nd]; st_val[sp] = v; st_len[sp] = (uint8_t)(l + 1); sp++;                    // left = 0
        st_node[sp] = right[nd]; st_val[sp] = v | (1u << l); st_len[sp] = (uint8_t)(l + 1); sp++;       // right = 1
    }
}

__global__ void __launch_bounds__(kEntThreads) k_ent_codes(EntFrame f, EntBufs b)
{
    __shared__ uint32_t val[16];
    __shared__ uint8_t len[16], table[16];
    const int stream = (int)blockIdx.x;
    EntCodes *out = b.codes + stream;
    if (threadIdx.x == 0) {
        int32_t hist[16];
        for (int i = 0; i < 16; i++) {
            hist[i] = b.hist[stream * 16 + i];
            b.hist[stream * 16 + i] = 0;   // ready for the next frame
        }
        uint32_t v[16];
        uint8_t l[16], t[16];
        ent_build_codes(hist, t, v, l);
        for (int i = 0; i < 16; i++) { val[i] = v[i]; len[i] = l[i]; table[i] = t[i]; }
    }
    __syncthreads();
    const unsigned z = threadIdx.x & 15u, nb = threadIdx.x >> 4;
    out->pair_bits[threadIdx.x] = val[z] | (val[nb] << len[z]);
    out->pair_len[threadIdx.x] = (uint8_t)(len[z] + len[nb]);
    if (threadIdx.x < 16) {
        out->len[threadIdx.x] = len[threadIdx.x];
        out->table[threadIdx.x] = table[threadIdx.x];
    }
    (void)f;
}

// ---------------------------------------------------------------------------------------------------- k_ent_offsets
// Exclusive prefix sum of `n` per-item bit counts, items dealt to the workgroup's threads in contiguous chunks.
// bits(i) is evaluated twice (count pass, write pass).  Returns the total.
template <class Bits>
__device__ inline uint32_t ent_block_scan(int n, uint32_t base, uint32_t *out, uint32_t *lds /*[kEntScanThreads/64 + 1]*/, Bits bits)
{
    const int per = (n + kEntScanThreads - 1) / kEntScanThreads;
    const int lo = (int)threadIdx.x * per, hi = min(lo + per, n);
    uint32_t mine = 0;
    for (int i = lo; i < hi; i++) mine += bits(i);
    // wavefront inclusive scan, then the wavefront totals
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = ent_shfl(incl, lane >= d ? lane - d : lane);
        if (lane >= d) incl += up;
    }
    __syncthreads();   // lds reuse across calls
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
    for (int w = 0; w < kEntScanThreads / 64; w++) {
        const uint32_t t = lds[w];
        if (w < wave) wave_base += t;
        total += t;
    }
    uint32_t run = base + wave_base + incl - mine;
    for (int i = lo; i < hi; i++) {
        out[i] = run;
        run += bits(i);
    }
    return total;
}

__global__ void __launch_bounds__(kEntScanThreads) k_ent_offsets(EntFrame f, EntBufs b)
{
    __shared__ uint32_t lds[kEntScanThreads / 64 + 1];
    __shared__ uint8_t len[16];
    const int stream = (int)blockIdx.x, n_sb = f.total_blocks * 4;
    const EntCodes *codes = b.codes + stream;
    if (threadIdx.x < 16) len[threadIdx.x] = codes->len[threadIdx.x];
    __syncthreads();
    uint32_t bit = 19u * 8u;   // 16 table bytes + 3 q-table indices (enc.rs:290-298, :403-411)
    if (f.pframe) {            // block headers: has_mvec, has_coeff, then two 7-bit components (enc.rs:414-451)
        const int8_t *mv = b.mv + (size_t)stream * f.total_blocks * 2;
        bit += ent_block_scan(f.total_blocks, bit, b.hdr_off + (size_t)stream * f.total_blocks, lds,
                              [&](int i) { return (mv[2 * i] != 0 || mv[2 * i + 1] != 0) ? 16u : 2u; });
    }
    const uint4 *counts = b.counts + (size_t)stream * n_sb;
    const uint32_t *sumsize = b.sumsize + (size_t)stream * n_sb;
    bit += ent_block_scan(n_sb, bit, b.sb_off + (size_t)stream * n_sb, lds, [&](int i) {
        const uint4 c = counts[i];
        const uint32_t w[4] = {c.x, c.y, c.z, c.w};
        uint32_t bits = sumsize[i];
#pragma unroll
        for (int k = 0; k < 16; k++) bits += ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) * len[k];
        return bits;
    });
    if (threadIdx.x == 0) {
        uint32_t bytes = (bit + 7u) >> 3;
        if (codes->oversize) bytes = kEntErrOversize;
        else if (bytes > f.cap_bytes) bytes = kEntErrCapacity;
        b.sizes[stream] = bytes;
    }
}

// ---------------------------------------------------------------------------------------------------- k_ent_init
__global__ void __launch_bounds__(kEntThreads) k_ent_init(EntFrame f, EntBufs b)
{
    const int stream = (int)blockIdx.y;
    const uint32_t bytes = b.sizes[stream];
    if (blockIdx.x == 0 && threadIdx.x == 0) b.codes[stream].oversize = 0;   // consumed by k_ent_offsets; next frame starts clean
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
// LSB-first bit writer of one lane: the first and the last word it touches may be shared with its neighbours in the
// stream (atomicOr onto the zeroed payload), every word in between is its own.
struct LaneBits {
    uint32_t *w;
    uint64_t acc = 0;
    unsigned fill;
    bool first = true;
    __device__ __forceinline__ LaneBits(uint32_t *words, uint32_t bit_off) : w(words + (bit_off >> 5)), fill(bit_off & 31u) {}
    __device__ __forceinline__ void put(uint32_t bits, unsigned len)   // len <= 30, bits < 2^len
    {
        acc |= (uint64_t)bits << fill;
        fill += len;
        if (fill >= 32u) {
            if (first) atomicOr(w, (uint32_t)acc);
            else *w = (uint32_t)acc;
            first = false;
            w++;
            acc >>= 32;
            fill -= 32u;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (fill && (uint32_t)acc) atomicOr(w, (uint32_t)acc);
    }
};

__global__ void __launch_bounds__(kEntThreads) k_ent_pack(EntFrame f, EntBufs b)
{
    __shared__ uint32_t pair_bits[256];
    __shared__ uint8_t pair_len[256];
    const int stream = (int)blockIdx.y, n_sb = f.total_blocks * 4;
    const EntCodes *codes = b.codes + stream;
    pair_bits[threadIdx.x] = codes->pair_bits[threadIdx.x];
    pair_len[threadIdx.x] = codes->pair_len[threadIdx.x];
    __syncthreads();
    const int sbi = (int)(blockIdx.x * kEntThreads + threadIdx.x);
    const bool ok = b.sizes[stream] < kEntErrCapacity;
    const bool live = sbi < n_sb && ok;
    const int mb = sbi >> 2, sb = sbi & 3;
    const size_t sbase = (size_t)stream * n_sb;
    uint32_t *words = (uint32_t *)(b.payload + (size_t)stream * f.cap_bytes);
    const uint64_t mask = live ? b.mask[sbase + sbi] : 0;
    int last = ent_prev_last(mask, sb);
    if (!live) return;

    if (f.pframe && sb == 0) {   // block header (enc.rs:414-451)
        const size_t bi = (size_t)stream * f.total_blocks + mb;
        const int mx = b.mv[2 * bi], my = b.mv[2 * bi + 1];
        const bool has_mvec = mx != 0 || my != 0;
        uint32_t bits = (has_mvec ? 1u : 0u) | (b.has[bi] ? 2u : 0u);
        unsigned len = 2;
        if (has_mvec) {
            bits |= ((uint32_t)mx & 0x7fu) << 2 | ((uint32_t)my & 0x7fu) << 9;
            len = 16;
        }
        LaneBits hw(words, b.hdr_off[bi]);
        hw.put(bits, len);
        hw.finish();
    }
    const bool coded = !f.pframe || b.has[(size_t)stream * f.total_blocks + mb] != 0;
    if (!coded) return;
    const int16_t *c = b.coef + ((size_t)stream * f.total_blocks + mb) * 256 + sb * 64;
    LaneBits bw(words, b.sb_off[sbase + sbi]);
    const uint32_t filler_bits = pair_bits[15], filler_len = pair_len[15];   // (15, size 0)
    for (uint64_t mm = mask; mm;) {
        const int bit = __builtin_ctzll(mm);
        mm &= mm - 1;
        const int i = 64 * sb + bit;
        unsigned fillers, rest;
        ent_split_run((unsigned)(i - last - 1), fillers, rest);
        last = i;
        const int v = c[bit];
        const unsigned mag = (unsigned)(v < 0 ? -v : v);
        const unsigned size = (33u - (unsigned)__builtin_clz(mag)) & 15u;
        for (; fillers; fillers--) bw.put(filler_bits, filler_len);
        const unsigned p = rest | (size << 4);
        bw.put(pair_bits[p], pair_len[p]);
        bw.put((uint32_t)v & ((1u << size) - 1u), size);   // write_signed: low `size` bits (enc.rs:313-315)
    }
    if (sb == 3 && last < 255) {
        unsigned fillers, rest;
        ent_split_run((unsigned)(255 - last), fillers, rest);
        for (; fillers; fillers--) bw.put(filler_bits, filler_len);
        bw.put(pair_bits[rest], pair_len[rest]);   // (rest, size 0)
    }
    bw.finish();
}

}  // namespace pfv
