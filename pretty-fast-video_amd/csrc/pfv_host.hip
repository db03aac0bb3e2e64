// pfv_host.hip -- host half of the drop-in: pfv::Encoder / pfv::Decoder session objects with the reference's
// surface (src/enc.rs:12-188, src/dec.rs:15-224), the .pfv container (src/enc.rs:190-235, src/dec.rs:38-118) and
// the host-side entropy layer (src/rle.rs, src/huffman.rs, src/enc.rs:237-481, src/dec.rs:226-448).  Entropy
// coding is serial, bit-granular work and stays on the host by design (BASELINE.json north_star); everything
// per-macroblock goes through the device sessions of pfv_capi.hip.  Included by pfv_capi.hip (one translation
// unit); exported through the extern "C" block at the bottom (include/pfv_hip.h).
//
// Bit I/O: the reference uses bitstream-io 1.6.0 BitWriter/BitReader<_, LittleEndian> (un-vendored crate): write(n, v)
// appends the low n bits of v LSB-first, write_signed(n, v) the n-bit two's complement of v LSB-first, byte_align
// pads with zeros.  Restated from the crate's documented behaviour; byte-level parity with a Rust-built stream is
// unpinned (no Rust toolchain, no real .pfv fixture: the ones in the reference are Git-LFS stubs).
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <emmintrin.h>
#include <memory>
#include <vector>

namespace pfv {

// ------------------------------------------------------------------ little-endian bit packing
class BitSink {
  public:
    explicit BitSink(std::vector<uint8_t> &out) : out_(out), len_(out.size()) {}
    void put(unsigned nbits, uint32_t value)   // BitWrite::write
    {
        if (nbits == 0) return;
        acc_ |= (uint64_t)(value & (nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u))) << fill_;
        fill_ += nbits;
        if (fill_ >= 32) {                     // flush four bytes at a time (little-endian host)
            room(4);
            const uint32_t lo = (uint32_t)acc_;
            std::memcpy(out_.data() + len_, &lo, 4);
            len_ += 4;
            acc_ >>= 32;
            fill_ -= 32;
        }
    }
    void put_signed(unsigned nbits, int32_t value) { put(nbits, (uint32_t)value); }   // BitWrite::write_signed (LE)
    void reserve(size_t more_bytes) { room(more_bytes + 8); }
    // put() without the capacity check: the caller reserved the bytes; value must already fit nbits (1..32)
    void put_reserved(unsigned nbits, uint32_t value)
    {
        acc_ |= (uint64_t)value << fill_;
        fill_ += nbits;
        if (fill_ >= 32) {
            const uint32_t lo = (uint32_t)acc_;
            std::memcpy(out_.data() + len_, &lo, 4);
            len_ += 4;
            acc_ >>= 32;
            fill_ -= 32;
        }
    }
    void align()                                                                        // BitWrite::byte_align
    {
        room(4);
        for (; fill_ > 0; fill_ = fill_ > 8 ? fill_ - 8 : 0, acc_ >>= 8) out_[len_++] = (uint8_t)acc_;
        acc_ = 0;
        out_.resize(len_);
    }

  private:
    void room(size_t n)
    {
        if (out_.size() < len_ + n) out_.resize(std::max(out_.size() * 2, len_ + n + 4096));
    }
    std::vector<uint8_t> &out_;
    size_t len_;
    uint64_t acc_ = 0;
    unsigned fill_ = 0;
};

class BitSource {
  public:
    BitSource(const uint8_t *p, size_t n) : p_(p), total_((uint64_t)n * 8) {}
    uint64_t total_bits() const { return total_; }
    uint64_t position() const { return pos_; }
    void set_position(uint64_t pos) { pos_ = pos; }
    const uint8_t *data() const { return p_; }
    void seek(int64_t delta) { pos_ = (uint64_t)((int64_t)pos_ + delta); }
    bool ok() const { return ok_; }
    uint32_t get(unsigned nbits)   // BitRead::read (nbits <= 32)
    {
        if (pos_ + nbits > total_) {
            ok_ = false;
            pos_ = total_;
            return 0;
        }
        if (nbits == 0) return 0;
        const size_t byte = (size_t)(pos_ >> 3), nbytes = (size_t)(total_ >> 3);
        uint64_t w = 0;
        if (byte + 8 <= nbytes) {
            std::memcpy(&w, p_ + byte, 8);                 // little-endian host: bit k of the stream = bit k of w
        } else {
            for (size_t k = 0; byte + k < nbytes && k < 8; k++) w |= (uint64_t)p_[byte + k] << (8 * k);
        }
        w >>= (pos_ & 7);
        pos_ += nbits;
        return (uint32_t)(w & (nbits >= 32 ? 0xffffffffull : ((1ull << nbits) - 1ull)));
    }
    // >= 57 valid stream bits starting at the read position, without consuming them; only when can_peek()
    bool can_peek() const { return pos_ + 72 <= total_; }
    uint64_t peek() const
    {
        uint64_t w;
        std::memcpy(&w, p_ + (size_t)(pos_ >> 3), 8);
        return w >> (pos_ & 7);
    }
    void skip(unsigned nbits) { pos_ += nbits; }
    int32_t get_signed(unsigned nbits)   // BitRead::read_signed (LE): n-bit two's complement
    {
        uint32_t v = get(nbits);
        if (nbits < 32 && ((v >> (nbits - 1)) & 1u)) v |= ~((1u << nbits) - 1u);
        return (int32_t)v;
    }

  private:
    const uint8_t *p_;
    uint64_t total_, pos_ = 0;
    bool ok_ = true;
};

// ------------------------------------------------------------------ src/huffman.rs
struct HuffCode {
    uint32_t val = 0, len = 0;
    uint8_t symbol = 0;
};

class HuffmanTree {
  public:
    // HuffmanTree::from_table (huffman.rs:71-119)
    explicit HuffmanTree(const std::array<uint8_t, 16> &table) : table_(table)
    {
        struct Item { uint32_t freq; int node; };
        std::vector<Item> list;
        for (int ch = 0; ch < 16; ch++)
            if (table[ch]) {
                nodes_.push_back({(int)ch, -1, -1});
                list.push_back({table[ch], (int)nodes_.size() - 1});
            }
        std::stable_sort(list.begin(), list.end(), [](const Item &a, const Item &b) { return a.freq > b.freq; });   // :81
        while (list.size() > 1) {
            Item a = list.back(); list.pop_back();
            Item b = list.back(); list.pop_back();
            nodes_.push_back({-1, a.node, b.node});                              // left = a, right = b (:87-88)
            Item c{a.freq + b.freq, (int)nodes_.size() - 1};
            auto pos = std::find_if(list.begin(), list.end(), [&](const Item &x) { return c.freq > x.freq; });   // :61-69
            list.insert(pos, c);
        }
        if (list.empty()) return;                                               // HuffmanTree::empty() (:95-97)
        root_ = list[0].node;
        assign(root_, HuffCode{});
        for (uint32_t val = 0; val < 256; val++)                                // fast table (:109-116)
            for (const HuffCode &c : codes_)
                if (c.len > 0 && c.len <= 8 && (val & ((1u << c.len) - 1u)) == c.val) {
                    fast_[val] = c;
                    break;
                }
    }
    const std::array<uint8_t, 16> &table() const { return table_; }
    const HuffCode &code(uint8_t sym) const { return codes_[sym & 15]; }
    const HuffCode &fast(uint32_t low8) const { return fast_[low8 & 255]; }   // len == 0: no code of <= 8 bits matches
    // (num_zeroes, coeff_size) decoded together from the low 12 bits of the window: used = total code bits (0: one of the
    // two codes is longer than 8 bits or the pair longer than 12 -- take the one-at-a-time path)
    struct PairEntry { uint8_t used, zeros, size; };
    const PairEntry &pair(uint32_t low12) const { return pair_[low12 & 4095]; }
    uint32_t pair32(uint32_t low12) const { return pair32_[low12 & 4095]; }   // the same entry in one word: used | zeros << 8 | size << 16
    void build_pair_table()   // once per packet that is worth it (the run parser of whole frames)
    {
        for (uint32_t v = 0; v < 4096; v++) {
            PairEntry e{0, 0, 0};
            const HuffCode &a = fast_[v & 255];
            if (a.len) {
                const HuffCode &b = fast_[(v >> a.len) & 255];
                if (b.len && a.len + b.len <= 12) e = PairEntry{(uint8_t)(a.len + b.len), a.symbol, b.symbol};
            }
            pair_[v] = e;
            pair32_[v] = (uint32_t)e.used | ((uint32_t)e.zeros << 8) | ((uint32_t)e.size << 16);
        }
    }

    // HuffmanTree::read (huffman.rs:156-197); -1 = DecodeError, -2 = I/O error
    int read(BitSource &r, uint64_t max_bits) const
    {
        uint64_t remaining = max_bits - r.position();
        unsigned nread = (unsigned)std::min<uint64_t>(remaining, 8);
        uint32_t cur = r.get(nread);
        if (!r.ok()) return -2;
        const HuffCode &c = fast_[cur & 255];
        if (c.len == 0) {
            r.seek(-(int64_t)nread);
            return read_slow(r);
        }
        r.seek(-((int64_t)nread - (int64_t)c.len));
        return c.symbol;
    }

  private:
    struct Node { int ch, left, right; };
    void assign(int n, HuffCode s)   // assign_codes (huffman.rs:204-217): left = 0, right = 1, first branch = LSB
    {
        const Node &nd = nodes_[n];
        if (nd.ch >= 0) {
            s.symbol = (uint8_t)nd.ch;
            codes_[nd.ch] = s;
            return;
        }
        HuffCode l = s, r = s;
        l.len = r.len = s.len + 1;
        r.val |= 1u << s.len;
        if (nd.left >= 0) assign(nd.left, l);
        if (nd.right >= 0) assign(nd.right, r);
    }
    int read_slow(BitSource &r) const   // huffman.rs:125-154
    {
        int n = root_;
        if (n < 0) return -1;
        while (nodes_[n].ch < 0) {
            uint32_t bit = r.get(1);
            if (!r.ok()) return -2;
            n = bit ? nodes_[n].right : nodes_[n].left;
            if (n < 0) return -1;
        }
        return nodes_[n].ch;
    }
    std::array<uint8_t, 16> table_;
    std::array<HuffCode, 16> codes_{};
    std::array<HuffCode, 256> fast_{};
    std::array<PairEntry, 4096> pair_{};
    std::array<uint32_t, 4096> pair32_{};
    std::vector<Node> nodes_;
    int root_ = -1;
};

// rle_create_huffman (rle.rs:49-66): histogram -> u8 table max(1, x*255/max)
inline std::array<uint8_t, 16> normalise_histogram(const std::array<int32_t, 16> &hist)
{
    int32_t mx = 0;
    for (int32_t x : hist) mx = std::max(mx, x);
    std::array<uint8_t, 16> t{};
    for (int i = 0; i < 16; i++)
        if (hist[i] > 0) t[i] = (uint8_t)std::max<int64_t>(1, (int64_t)hist[i] * 255 / mx);
    return t;
}

// ------------------------------------------------------------------ packet payloads
// Run symbols of all coded macroblocks back to back, one 32-bit word each: num_zeroes | coeff_size << 4 | coeff << 16
// (rle_encode + update_table, src/rle.rs:9-47, fused into one scan; one run stream per macroblock, enc.rs:246-255).
struct BlockRuns {
    const uint32_t *symbols = nullptr;   // into a per-thread scratch that only ever grows (no per-frame allocation)
    size_t count = 0;
    std::array<int32_t, 16> hist{};
};

inline bool collect_runs(BlockRuns &br, const int16_t *coef, const uint8_t *has /*nullable: all coded*/, int total_blocks)
{
    size_t coded = 0;
    for (int b = 0; b < total_blocks; b++) coded += (!has || has[b]) ? 1 : 0;
    static thread_local std::vector<uint32_t> scratch;
    const size_t need = coded * (256 + 18);   // worst case per macroblock: 256 values (or 17 fillers + 1 tail run)
    if (scratch.size() < need) scratch.resize(need);
    uint32_t *out = scratch.data();
    int32_t *hist = br.hist.data();
    const __m128i zero = _mm_setzero_si128();
    for (int b = 0; b < total_blocks; b++) {
        if (has && !has[b]) continue;
        const int16_t *d = coef + (size_t)b * 256;
        int last = -1;   // index of the previous non-zero coefficient
        for (int base = 0; base < 256; base += 32) {
            // one bit per coefficient: 1 = non-zero (SSE2 is baseline x86-64)
            uint32_t nz = 0;
            for (int k = 0; k < 4; k++) {
                const __m128i v = _mm_loadu_si128((const __m128i *)(d + base + 8 * k));
                const uint32_t eq = (uint32_t)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(v, zero), zero)) & 0xffu;
                nz |= (eq ^ 0xffu) << (8 * k);
            }
            while (nz) {
                const int i = base + __builtin_ctz(nz);
                nz &= nz - 1;
                unsigned run = (unsigned)(i - last - 1);
                last = i;
                for (; run > 15; run -= 15) { *out++ = 15u; hist[15]++; hist[0]++; }       // (15, size 0) fillers (rle.rs:18-21)
                const int v = d[i];
                const unsigned mag = (unsigned)(v < 0 ? -v : v);
                const unsigned size = 33u - (unsigned)__builtin_clz(mag);                   // bit length + 1 (rle.rs:23-24)
                if (size > 15) return false;                                                // the reference would index hist[16+]
                *out++ = run | (size << 4) | ((uint32_t)(uint16_t)v << 16);
                hist[run]++;
                hist[size]++;
            }
        }
        unsigned run = (unsigned)(255 - last);
        for (; run > 15; run -= 15) { *out++ = 15u; hist[15]++; hist[0]++; }               // rle.rs:31-34
        if (run) { *out++ = run; hist[run]++; hist[0]++; }                                  // trailing run (rle.rs:36-38)
    }
    br.symbols = scratch.data();
    br.count = (size_t)(out - scratch.data());
    return true;
}
inline void emit_runs(std::vector<uint8_t> &payload, BitSink &w, const HuffmanTree &tree, const BlockRuns &br)
{
    (void)payload;
    // (num_zeroes, coeff_size) code pairs pre-joined: two tree codes of <= 15 bits each fit one 30-bit put
    struct Pair { uint32_t bits; uint32_t len; };
    Pair pair[256];
    for (unsigned z = 0; z < 16; z++)
        for (unsigned n = 0; n < 16; n++) {
            const HuffCode &cz = tree.code((uint8_t)z), &cn = tree.code((uint8_t)n);
            pair[z | (n << 4)] = {cz.val | (cn.val << cz.len), cz.len + cn.len};
        }
    w.reserve(br.count * 6);   // <= 30 + 15 bits per symbol
    for (size_t k = 0; k < br.count; k++) {
        const uint32_t s = br.symbols[k];
        const Pair &p = pair[s & 255u];
        if (p.len) w.put_reserved(p.len, p.bits);
        const unsigned size = (s >> 4) & 15u;
        // write_signed: the low `size` bits of the two's complement (enc.rs:313-315)
        if (size) w.put_reserved(size, (s >> 16) & ((1u << size) - 1u));
    }
}

// write_iframe_packet payload (enc.rs:237-320)
inline bool serialize_iframe(std::vector<uint8_t> &payload, const int16_t *coef, int total_blocks)
{
    BlockRuns br;
    if (!collect_runs(br, coef, nullptr, total_blocks)) return false;
    HuffmanTree tree(normalise_histogram(br.hist));
    BitSink w(payload);
    for (uint8_t t : tree.table()) w.put(8, t);
    w.put(8, 0); w.put(8, 1); w.put(8, 1);   // q-table index per plane: intra_l, intra_c, intra_c (enc.rs:296-298)
    emit_runs(payload, w, tree, br);
    w.align();
    return true;
}
// write_pframe_packet payload (enc.rs:332-470)
inline bool serialize_pframe(std::vector<uint8_t> &payload, const int8_t *mv, const uint8_t *has, const int16_t *coef,
                             int total_blocks)
{
    BlockRuns br;
    if (!collect_runs(br, coef, has, total_blocks)) return false;
    HuffmanTree tree(normalise_histogram(br.hist));
    BitSink w(payload);
    for (uint8_t t : tree.table()) w.put(8, t);
    w.put(8, 2); w.put(8, 3); w.put(8, 3);   // inter_l, inter_c, inter_c (enc.rs:409-411)
    for (int b = 0; b < total_blocks; b++) { // block headers, Y then U then V (enc.rs:414-451)
        bool has_mvec = mv[2 * b] != 0 || mv[2 * b + 1] != 0;
        w.put(1, has_mvec);
        w.put(1, has[b] != 0);
        if (has_mvec) {
            w.put_signed(7, mv[2 * b]);
            w.put_signed(7, mv[2 * b + 1]);
        }
    }
    emit_runs(payload, w, tree, br);
    w.align();
    return true;
}

// Packet payload parsers (dec.rs:226-296, 328-417).  0 = ok, PFV_ERR_* otherwise.
struct PacketHead {
    std::array<uint8_t, 16> table;
    uint8_t qidx[3];
};
inline int parse_head(BitSource &r, PacketHead &h, int n_qtables)
{
    for (auto &t : h.table) t = (uint8_t)r.get(8);
    for (auto &q : h.qidx) q = (uint8_t)r.get(8);
    if (!r.ok()) return -8;
    // the reference indexes self.qtables the moment it reads each index (dec.rs:244-246, 346-348): an index past the
    // header's table count fails here, before any run is parsed
    for (uint8_t q : h.qidx)
        if (q >= n_qtables) return -6;
    return 0;
}
// Where parsed coefficients go: the dense [macroblock][256] array, or a list of (flat index, value) pairs -- after
// quantisation ~9 in 10 coefficients are zero, so the list is what is worth sending over PCIe (pfv_dec_*_sparse).
struct DenseSink {
    int16_t *out;
    bool put(size_t i, int16_t v) { out[i] = v; return true; }
    bool put_if(size_t c, size_t i, int16_t v) { if (c) out[i] = v; return true; }
};
struct SparseSink {
    uint32_t *idx;
    int16_t *val;
    size_t cap, n = 0;
    size_t offset = 0;   // added to every index: the frame's position in a wider [stream][macroblock][256] array
    bool put(size_t i, int16_t v)
    {
        if (n >= cap) return false;
        idx[n] = (uint32_t)(i + offset);
        val[n++] = v;
        return true;
    }
    // put(i, v) when c, nothing otherwise -- without a branch on c: the slot is written either way and only claimed when c
    bool put_if(size_t c, size_t i, int16_t v)
    {
        if (__builtin_expect(n >= cap, 0)) return !c;
        idx[n] = (uint32_t)(i + offset);
        val[n] = v;
        n += c;
        return true;
    }
};
// Coefficient lists (pfv_device.h: CoefLists), the form the decode kernels expand in LDS: one 32-bit entry per value and, per macroblock
// (and one more behind the last), the number of entries before it.  finish() writes the counts behind the last value.  The parsers hand
// over ascending indices (the run streams are read front to back).
struct ListSink {
    uint32_t *ent;
    size_t cap;
    uint32_t *counts;              // [total_blocks + 1]
    size_t total_blocks;
    size_t n = 0;
    size_t next_mb = 0;            // counts[0 .. next_mb) are written
    bool put(size_t i, int16_t v)
    {
        if (n >= cap) return false;
        const size_t mb = i >> 8;
        for (; next_mb <= mb; next_mb++) counts[next_mb] = (uint32_t)n;
        ent[n++] = coef_entry((uint32_t)mb, (uint32_t)i & 255u, v);
        return true;
    }
    bool put_if(size_t c, size_t i, int16_t v) { return c ? put(i, v) : true; }
    void finish()
    {
        for (; next_mb <= total_blocks; next_mb++) counts[next_mb] = (uint32_t)n;
    }
};
constexpr int kSinkFull = 1;   // SparseSink / ListSink ran out of room: the caller falls back (dense form; a list of the frame's full size)

// reads run symbols until `count` coefficients are covered, handing values to the sink (base + index, value).
// Fast section (while at least kFastTailBits of stream lie behind the read position): a 64-bit bit buffer refilled without a branch
// and OFF the decode's dependency chain (buf |= load(ptr) << have; ptr += (63 - have) >> 3; have |= 56 -- the bytes that overlap the
// buffer's top are the same bits again), two symbols per refill (a symbol is at most 12 code bits, the pair table, + 15 value bits),
// value / no value handled without a branch: half of all symbols carry no value (fillers, closing runs), unpredictably mixed with those
// that do, and those mispredictions were a third of the parser's time.  What the reference does per symbol is unchanged
// (huffman.rs:156-197, dec.rs:261-296, 378-417); the last bytes of a packet and codes longer than the table go the one-field-at-a-time way.
constexpr uint64_t kFastTailBits = 192;
template <class Sink>
inline int read_runs(BitSource &r, const HuffmanTree &tree, Sink &sink, size_t base, size_t count)
{
    size_t idx = 0;
    const uint64_t total = r.total_bits();
    while (idx < count) {
        const uint64_t pos = r.position();
        if (pos + kFastTailBits <= total) {
            const uint8_t *const data = r.data();
            const uint8_t *ptr = data + (pos >> 3);
            uint64_t buf;
            std::memcpy(&buf, ptr, 8);
            ptr += 7;
            unsigned have = 56 - (unsigned)(pos & 7);          // valid bits claimed in buf (the byte above them is re-read by the refill)
            buf >>= (pos & 7);
            const uint64_t stop = total - kFastTailBits;
            bool slow = false;
            for (;;) {
                uint64_t w;
                std::memcpy(&w, ptr, 8);
                buf |= w << have;
                ptr += (63 - have) >> 3;
                have |= 56;                                    // 56..63 valid bits: two symbols of at most 27 each
#define PFV_RUN_SYMBOL()                                                                                                  \
                {                                                                                                         \
                    const uint32_t pe = tree.pair32((uint32_t)buf);          /* used | zeros << 8 | size << 16 */        \
                    if (!(pe & 0xffu)) { slow = true; break; }                                                            \
                    const unsigned pused = pe & 0xffu, nb = pe >> 16;                                                     \
                    idx += (pe >> 8) & 0xffu;                                                                             \
                    if (idx >= count) {             /* the run closes the macroblock, or the stream is damaged */         \
                        if (nb) return -6;          /* the reference would index out of bounds here */                    \
                        buf >>= pused;                                                                                    \
                        have -= pused;                                                                                    \
                        break;                                                                                            \
                    }                                                                                                     \
                    const uint32_t sign = (1u << nb) >> 1;                                  /* 0 for nb == 0 */           \
                    const uint32_t raw = (uint32_t)(buf >> pused) & ((1u << nb) - 1u);                                    \
                    const int16_t v = (int16_t)(int32_t)((raw ^ sign) - sign);              /* sign-extend nb bits */     \
                    const size_t has_value = nb != 0;                                                                     \
                    if (!sink.put_if(has_value, base + idx, v)) return kSinkFull;                                         \
                    idx += has_value;                                                                                     \
                    const unsigned used = pused + nb;                                                                     \
                    buf >>= used;                                                                                         \
                    have -= used;                                                                                         \
                    if (idx >= count) break;                                                                              \
                }
                PFV_RUN_SYMBOL()
                PFV_RUN_SYMBOL()
#undef PFV_RUN_SYMBOL
                if ((uint64_t)(ptr - data) * 8 - have > stop) break;
            }
            r.set_position((uint64_t)(ptr - data) * 8 - have);
            if (!slow) continue;
        }
        int z = tree.read(r, r.total_bits());
        if (z < 0) return z == -2 ? -8 : -6;
        idx += (size_t)z;
        int nb = tree.read(r, r.total_bits());
        if (nb < 0) return nb == -2 ? -8 : -6;
        if (nb > 0) {   // nb == 0: a pure run of zeros (dec.rs:285)
            int32_t c = r.get_signed((unsigned)nb);
            if (!r.ok()) return -8;
            if (idx >= count) return -6;   // the reference would index out of bounds here
            if (!sink.put(base + idx++, (int16_t)c)) return kSinkFull;
        }
    }
    return 0;
}
template <class Sink>
inline int parse_iframe_to(const uint8_t *payload, size_t n, int total_blocks, int n_qtables, Sink &sink, uint8_t qidx[3])
{
    BitSource r(payload, n);
    PacketHead h;
    if (int rc = parse_head(r, h, n_qtables)) return rc;
    HuffmanTree tree(h.table);
    tree.build_pair_table();
    std::memcpy(qidx, h.qidx, 3);
    return read_runs(r, tree, sink, 0, (size_t)total_blocks * 256);   // ONE run stream for the whole frame (dec.rs:261)
}
// the block headers of a p-frame payload (dec.rs:361-372): [has_mvec][has_coeff]([mx:7s][my:7s]); returns the number of coded macroblocks
inline size_t parse_block_headers(BitSource &r, int total_blocks, int8_t *mv, uint8_t *has, uint32_t *coded = nullptr /* [total_blocks]: the coded macroblocks, in order */)
{
    size_t n_coded = 0;
    uint32_t scratch = 0;
    const size_t step = coded ? 1 : 0;
    if (!coded) coded = &scratch;
    for (int b = 0; b < total_blocks; b++) {
        if (r.can_peek()) {   // the whole block header (2 or 16 bits) from one window
            const uint32_t w = (uint32_t)r.peek();
            has[b] = (uint8_t)((w >> 1) & 1u);
            if (w & 1u) {
                mv[2 * b] = (int8_t)((int32_t)(w << 23) >> 25);       // bits 2..8, two's complement
                mv[2 * b + 1] = (int8_t)((int32_t)(w << 16) >> 25);   // bits 9..15
                r.skip(16);
            } else {
                mv[2 * b] = mv[2 * b + 1] = 0;
                r.skip(2);
            }
            coded[n_coded * step] = (uint32_t)b;     // claimed only when the macroblock is coded
            n_coded += has[b];
            continue;
        }
        bool has_mvec = r.get(1) != 0;
        has[b] = (uint8_t)r.get(1);
        mv[2 * b] = mv[2 * b + 1] = 0;
        if (has_mvec) {
            mv[2 * b] = (int8_t)r.get_signed(7);
            mv[2 * b + 1] = (int8_t)r.get_signed(7);
        }
        coded[n_coded * step] = (uint32_t)b;
        n_coded += has[b];
    }
    return n_coded;
}
template <class Sink>
inline int parse_pframe_to(const uint8_t *payload, size_t n, int total_blocks, int n_qtables, int8_t *mv, uint8_t *has,
                           Sink &sink, uint8_t qidx[3])
{
    BitSource r(payload, n);
    PacketHead h;
    if (int rc = parse_head(r, h, n_qtables)) return rc;
    HuffmanTree tree(h.table);
    tree.build_pair_table();
    std::memcpy(qidx, h.qidx, 3);
    (void)parse_block_headers(r, total_blocks, mv, has);
    if (!r.ok()) return -8;
    for (int b = 0; b < total_blocks; b++)     // dec.rs:378-417: 256 coefficients per coded macroblock
        if (has[b])
            if (int rc = read_runs(r, tree, sink, (size_t)b * 256, 256)) return rc;
    return 0;
}
inline int parse_iframe(const uint8_t *payload, size_t n, int total_blocks, int n_qtables, int16_t *coef, uint8_t qidx[3])
{
    std::memset(coef, 0, (size_t)total_blocks * 512);
    DenseSink sink{coef};
    return parse_iframe_to(payload, n, total_blocks, n_qtables, sink, qidx);
}
inline int parse_pframe(const uint8_t *payload, size_t n, int total_blocks, int n_qtables, int8_t *mv, uint8_t *has,
                        int16_t *coef, uint8_t qidx[3])
{
    std::memset(coef, 0, (size_t)total_blocks * 512);
    DenseSink sink{coef};
    return parse_pframe_to(payload, n, total_blocks, n_qtables, mv, has, sink, qidx);
}

}  // namespace pfv
