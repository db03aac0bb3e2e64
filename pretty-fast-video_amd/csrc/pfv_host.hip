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
#include <memory>
#include <vector>

namespace pfv {

// ------------------------------------------------------------------ little-endian bit packing
class BitSink {
  public:
    explicit BitSink(std::vector<uint8_t> &out) : out_(out) {}
    void put(unsigned nbits, uint32_t value)   // BitWrite::write
    {
        if (nbits == 0) return;
        acc_ |= (uint64_t)(value & (nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u))) << fill_;
        fill_ += nbits;
        while (fill_ >= 8) {
            out_.push_back((uint8_t)acc_);
            acc_ >>= 8;
            fill_ -= 8;
        }
    }
    void put_signed(unsigned nbits, int32_t value) { put(nbits, (uint32_t)value); }   // BitWrite::write_signed (LE)
    void align()                                                                        // BitWrite::byte_align
    {
        if (fill_) {
            out_.push_back((uint8_t)acc_);
            acc_ = 0;
            fill_ = 0;
        }
    }

  private:
    std::vector<uint8_t> &out_;
    uint64_t acc_ = 0;
    unsigned fill_ = 0;
};

class BitSource {
  public:
    BitSource(const uint8_t *p, size_t n) : p_(p), total_((uint64_t)n * 8) {}
    uint64_t total_bits() const { return total_; }
    uint64_t position() const { return pos_; }
    void seek(int64_t delta) { pos_ = (uint64_t)((int64_t)pos_ + delta); }
    bool ok() const { return ok_; }
    uint32_t get(unsigned nbits)   // BitRead::read
    {
        uint32_t v = 0;
        if (pos_ + nbits > total_) {
            ok_ = false;
            pos_ = total_;
            return 0;
        }
        for (unsigned got = 0; got < nbits;) {
            unsigned off = (unsigned)(pos_ & 7), take = std::min(8u - off, nbits - got);
            v |= (uint32_t)((p_[pos_ >> 3] >> off) & ((1u << take) - 1u)) << got;
            got += take;
            pos_ += take;
        }
        return v;
    }
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

// ------------------------------------------------------------------ src/rle.rs
struct RunSymbol {   // RLESequence (rle.rs:3-7)
    uint8_t num_zeroes, coeff_size;
    int16_t coeff;
};

// rle_encode (rle.rs:9-39).  Returns false if a coefficient needs more than 15 size bits (the reference's
// update_table would index its 16-entry histogram out of range, rle.rs:44).
inline bool rle_encode(std::vector<RunSymbol> &into, const int16_t *data, size_t n)
{
    unsigned run = 0;
    auto flush_long_run = [&]() {
        for (; run > 15; run -= 15) into.push_back({15, 0, 0});
    };
    for (size_t i = 0; i < n; i++) {
        const int v = data[i];
        if (v == 0) {
            run++;
            continue;
        }
        flush_long_run();
        unsigned mag = (unsigned)(v < 0 ? -v : v), bits = 0;
        while (mag >> bits) bits++;
        if (bits + 1 > 15) return false;
        into.push_back({(uint8_t)run, (uint8_t)(bits + 1), (int16_t)v});
        run = 0;
    }
    flush_long_run();
    if (run) into.push_back({(uint8_t)run, 0, 0});
    return true;
}

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
struct BlockRuns {
    std::vector<RunSymbol> symbols;   // all coded macroblocks back to back
    std::array<int32_t, 16> hist{};
};

inline bool collect_runs(BlockRuns &br, const int16_t *coef, const uint8_t *has /*nullable: all coded*/, int total_blocks)
{
    for (int b = 0; b < total_blocks; b++) {
        if (has && !has[b]) continue;
        size_t first = br.symbols.size();
        if (!rle_encode(br.symbols, coef + (size_t)b * 256, 256)) return false;   // one run stream per macroblock (enc.rs:246-255)
        for (size_t i = first; i < br.symbols.size(); i++) {                     // update_table (rle.rs:41-47)
            br.hist[br.symbols[i].num_zeroes]++;
            br.hist[br.symbols[i].coeff_size]++;
        }
    }
    return true;
}
inline void emit_runs(BitSink &w, const HuffmanTree &tree, const BlockRuns &br)
{
    for (const RunSymbol &s : br.symbols) {
        const HuffCode &z = tree.code(s.num_zeroes), &n = tree.code(s.coeff_size);
        w.put(z.len, z.val);
        w.put(n.len, n.val);
        if (s.coeff_size) w.put_signed(s.coeff_size, s.coeff);
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
    emit_runs(w, tree, br);
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
    emit_runs(w, tree, br);
    w.align();
    return true;
}

// Packet payload parsers (dec.rs:226-296, 328-417).  0 = ok, PFV_ERR_* otherwise.
struct PacketHead {
    std::array<uint8_t, 16> table;
    uint8_t qidx[3];
};
inline int parse_head(BitSource &r, PacketHead &h)
{
    for (auto &t : h.table) t = (uint8_t)r.get(8);
    for (auto &q : h.qidx) q = (uint8_t)r.get(8);
    return r.ok() ? 0 : -8;
}
// reads run symbols until `count` coefficients are covered, writing into out[0..count)
inline int read_runs(BitSource &r, const HuffmanTree &tree, int16_t *out, size_t count)
{
    size_t idx = 0;
    while (idx < count) {
        int z = tree.read(r, r.total_bits());
        if (z < 0) return z == -2 ? -8 : -6;
        idx += (size_t)z;
        int nb = tree.read(r, r.total_bits());
        if (nb < 0) return nb == -2 ? -8 : -6;
        if (nb > 0) {   // nb == 0: a pure run of zeros (dec.rs:285)
            int32_t c = r.get_signed((unsigned)nb);
            if (!r.ok()) return -8;
            if (idx >= count) return -6;   // the reference would index out of bounds here
            out[idx++] = (int16_t)c;
        }
    }
    return 0;
}
inline int parse_iframe(const uint8_t *payload, size_t n, int total_blocks, int16_t *coef, uint8_t qidx[3])
{
    BitSource r(payload, n);
    PacketHead h;
    if (int rc = parse_head(r, h)) return rc;
    HuffmanTree tree(h.table);
    std::memcpy(qidx, h.qidx, 3);
    std::memset(coef, 0, (size_t)total_blocks * 512);
    return read_runs(r, tree, coef, (size_t)total_blocks * 256);   // ONE run stream for the whole frame (dec.rs:261)
}
inline int parse_pframe(const uint8_t *payload, size_t n, int total_blocks, int8_t *mv, uint8_t *has, int16_t *coef,
                        uint8_t qidx[3])
{
    BitSource r(payload, n);
    PacketHead h;
    if (int rc = parse_head(r, h)) return rc;
    HuffmanTree tree(h.table);
    std::memcpy(qidx, h.qidx, 3);
    for (int b = 0; b < total_blocks; b++) {   // dec.rs:361-372
        bool has_mvec = r.get(1) != 0;
        has[b] = (uint8_t)r.get(1);
        mv[2 * b] = mv[2 * b + 1] = 0;
        if (has_mvec) {
            mv[2 * b] = (int8_t)r.get_signed(7);
            mv[2 * b + 1] = (int8_t)r.get_signed(7);
        }
    }
    if (!r.ok()) return -8;
    std::memset(coef, 0, (size_t)total_blocks * 512);
    for (int b = 0; b < total_blocks; b++)     // dec.rs:378-417: 256 coefficients per coded macroblock
        if (has[b])
            if (int rc = read_runs(r, tree, coef + (size_t)b * 256, 256)) return rc;
    return 0;
}

}  // namespace pfv
