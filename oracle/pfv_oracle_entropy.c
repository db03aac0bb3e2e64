/*
 * pfv_oracle_entropy.c -- CPU restatement of the pfv-rs host bitstream layer:
 * RLE (src/rle.rs), 16-symbol Huffman (src/huffman.rs), packet writers (src/enc.rs:190-481), packet parsers and
 * the Decoder packet loop (src/dec.rs:38-448).
 *
 * TEST INFRASTRUCTURE, like pfv_oracle.c: loaded only by tests/, smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: the bit-level I/O of the reference lives in the un-vendored crate bitstream-io 1.6.0
 * (Cargo.toml:24), whose source is not in /root/reference.  Its published LittleEndian semantics are restated
 * here: write(bits, v) appends the low `bits` bits of v LSB-first; write_signed(bits, v) appends the `bits`-wide
 * two's-complement value LSB-first (low bits-1 bits, then the sign bit); byte_align pads with zero bits.  The
 * reference's own tests at this boundary are round trips only (src/lib.rs:96-239; the fixture of the second one
 * is an LFS stub), so byte-level parity of the .pfv stream is UNPINNED by anything runnable; what is checked is
 * self-consistency, the inline vector of src/lib.rs:98, and agreement with the independently written product
 * implementation (pretty-fast-video_amd/csrc/pfv_host.hip).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PFVO_API __attribute__((visibility("default")))

/* from pfv_oracle.c */
typedef struct pfvo_encoder pfvo_encoder;
typedef struct pfvo_decoder pfvo_decoder;
pfvo_encoder *pfvo_encoder_new(int width, int height, int quality, int threads);
void pfvo_encoder_free(pfvo_encoder *e);
int pfvo_encoder_total_blocks(const pfvo_encoder *e);
void pfvo_encode_iframe(pfvo_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int16_t *coef_out);
void pfvo_encode_pframe(pfvo_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v, int8_t *mv_out,
                        uint8_t *has_coef_out, int16_t *coef_out);
void pfvo_qtables(int quality, int32_t intra_l[64], int32_t intra_c[64], int32_t inter_l[64], int32_t inter_c[64],
                  float *px_err);
pfvo_decoder *pfvo_decoder_new(int width, int height, const int32_t *qtables, int n_qtables, int threads);
void pfvo_decoder_free(pfvo_decoder *d);
const uint8_t *pfvo_decoder_plane(const pfvo_decoder *d, int p, int *pw, int *ph);
void pfvo_decode_iframe(pfvo_decoder *d, const int16_t *coef, const uint8_t qidx[3]);
int pfvo_decode_pframe(pfvo_decoder *d, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef, const uint8_t qidx[3]);
int pfvo_pad16(int x);

/* ---------------------------------------------------------------- growable byte buffer + LE bit writer */
typedef struct { uint8_t *p; size_t n, cap; } bytes;
static void b_put(bytes *b, const void *src, size_t n)
{
    if (b->n + n > b->cap) {
        size_t c = b->cap ? b->cap : 4096;
        while (c < b->n + n) c *= 2;
        b->p = (uint8_t *)realloc(b->p, c);
        b->cap = c;
    }
    memcpy(b->p + b->n, src, n);
    b->n += n;
}
static void b_u8(bytes *b, uint8_t v) { b_put(b, &v, 1); }
static void b_u16(bytes *b, uint16_t v) { uint8_t t[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; b_put(b, t, 2); }
static void b_u32(bytes *b, uint32_t v) { uint8_t t[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; b_put(b, t, 4); }

typedef struct { bytes *out; uint32_t acc; int nbits; } bitw;   /* bitstream-io BitWriter<_, LittleEndian> */
static void bw_write(bitw *w, int bits, uint32_t val)
{
    for (int i = 0; i < bits; i++) {
        w->acc |= ((val >> i) & 1u) << w->nbits;
        if (++w->nbits == 8) { b_u8(w->out, (uint8_t)w->acc); w->acc = 0; w->nbits = 0; }
    }
}
static void bw_write_signed(bitw *w, int bits, int32_t v) { bw_write(w, bits, (uint32_t)v & ((bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u))); }
static void bw_align(bitw *w) { if (w->nbits) { b_u8(w->out, (uint8_t)w->acc); w->acc = 0; w->nbits = 0; } }

/* bitstream-io BitReader<Cursor<&[u8]>, LittleEndian> with bit-granular position */
typedef struct { const uint8_t *p; uint64_t nbits, pos; int err; } bitr;
static uint32_t br_read(bitr *r, int bits)
{
    uint32_t v = 0;
    for (int i = 0; i < bits; i++) {
        if (r->pos >= r->nbits) { r->err = 1; return 0; }
        v |= (uint32_t)((r->p[r->pos >> 3] >> (r->pos & 7)) & 1u) << i;
        r->pos++;
    }
    return v;
}
static int32_t br_read_signed(bitr *r, int bits)
{
    uint32_t v = br_read(r, bits);
    if (bits < 32 && (v & (1u << (bits - 1)))) v |= ~((1u << bits) - 1u);
    return (int32_t)v;
}

/* ---------------------------------------------------------------- src/rle.rs */
typedef struct { uint8_t num_zeroes, coeff_size; int16_t coeff; } rle_seq;   /* rle.rs:3-7 */
typedef struct { rle_seq *p; size_t n, cap; } rle_vec;
static void rv_push(rle_vec *v, uint8_t z, uint8_t s, int16_t c)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = (rle_seq *)realloc(v->p, v->cap * sizeof(rle_seq)); }
    v->p[v->n].num_zeroes = z; v->p[v->n].coeff_size = s; v->p[v->n].coeff = c; v->n++;
}
/* rle.rs:9-39 */
static void rle_encode(rle_vec *into, const int16_t *data, size_t len)
{
    uint32_t run = 0;
    for (size_t idx = 0; idx < len; idx++) {
        int16_t val = data[idx];
        if (val == 0) { run++; continue; }
        while (run > 15) { rv_push(into, 15, 0, 0); run -= 15; }                 /* :18-21 */
        uint16_t c = (uint16_t)(val < 0 ? -val : val);                            /* val.abs() as u16 (:23) */
        int lz = 16; for (uint16_t t = c; t; t >>= 1) lz--;
        int numbits = (16 - lz) + 1;                                              /* :24 */
        rv_push(into, (uint8_t)run, (uint8_t)numbits, val);
        run = 0;
    }
    while (run > 15) { rv_push(into, 15, 0, 0); run -= 15; }                     /* :31-34 */
    if (run > 0) rv_push(into, (uint8_t)run, 0, 0);                               /* :36-38 */
}
/* rle.rs:41-47 */
static void update_table(int32_t table[16], const rle_seq *s, size_t n)
{
    for (size_t i = 0; i < n; i++) { table[s[i].num_zeroes & 15]++; table[s[i].coeff_size & 15]++; }
}

/* ---------------------------------------------------------------- src/huffman.rs */
typedef struct { uint32_t val, len; uint8_t symbol; } hcode;                      /* huffman.rs:18-22 */
typedef struct hnode { uint32_t freq; int ch; int left, right; } hnode;           /* :39-44; children as indices */
typedef struct {
    hcode codes[16]; uint8_t table[16]; hcode dec_table[256];
    hnode nodes[32]; int root; int empty;
} htree;

static void assign_codes(const htree *t, int n, hcode *h, hcode s)                /* :204-217 */
{
    const hnode *p = &t->nodes[n];
    if (p->ch >= 0) { s.symbol = (uint8_t)p->ch; h[p->ch] = s; return; }
    if (p->left >= 0) { hcode l = s; l.len = s.len + 1; assign_codes(t, p->left, h, l); }                       /* append(false) */
    if (p->right >= 0) { hcode r = s; r.val = s.val | (1u << s.len); r.len = s.len + 1; assign_codes(t, p->right, h, r); }
}
/* huffman.rs:71-119 */
static void huff_from_table(htree *t, const uint8_t table[16])
{
    memset(t, 0, sizeof *t);
    memcpy(t->table, table, 16);
    int p[16], np = 0, nn = 0;
    for (int ch = 0; ch < 16; ch++)
        if (table[ch] > 0) { t->nodes[nn].freq = table[ch]; t->nodes[nn].ch = ch; t->nodes[nn].left = t->nodes[nn].right = -1; p[np++] = nn++; }
    /* stable sort, descending frequency (:81) */
    for (int i = 1; i < np; i++) {
        int x = p[i], j = i - 1;
        while (j >= 0 && t->nodes[p[j]].freq < t->nodes[x].freq) { p[j + 1] = p[j]; j--; }
        p[j + 1] = x;
    }
    while (np > 1) {
        int a = p[--np], b = p[--np];                                             /* :84-85 */
        int c = nn++;
        t->nodes[c].freq = t->nodes[a].freq + t->nodes[b].freq; t->nodes[c].ch = -1;
        t->nodes[c].left = a; t->nodes[c].right = b;                              /* :87-88 */
        int pos = np;                                                             /* get_insert_index :61-69 */
        for (int i = 0; i < np; i++) if (t->nodes[c].freq > t->nodes[p[i]].freq) { pos = i; break; }
        for (int i = np; i > pos; i--) p[i] = p[i - 1];
        p[pos] = c; np++;
    }
    if (np == 0) { t->empty = 1; t->root = -1; return; }                          /* :95-97 HuffmanTree::empty() */
    t->root = p[0];
    hcode s0 = {0, 0, 0};
    assign_codes(t, t->root, t->codes, s0);
    for (uint32_t val = 0; val < 256; val++)                                      /* :109-116 */
        for (int k = 0; k < 16; k++) {
            hcode c = t->codes[k];
            if (c.len > 0 && c.len <= 8 && (val & ((1u << c.len) - 1u)) == c.val) { t->dec_table[val] = c; break; }
        }
}
/* rle.rs:49-66 */
static void rle_create_huffman(htree *t, const int32_t table[16])
{
    int32_t max = 0;
    for (int i = 0; i < 16; i++) if (table[i] > max) max = table[i];
    uint8_t tb[16];
    for (int i = 0; i < 16; i++) {
        if (table[i] > 0) { int32_t v = (int32_t)(((int64_t)table[i] * 255) / max); tb[i] = (uint8_t)(v < 1 ? 1 : v); }
        else tb[i] = 0;
    }
    huff_from_table(t, tb);
}
/* huffman.rs:125-154 read_slow; returns -1 on DecodeError, -2 on IO error */
static int huff_read_slow(const htree *t, bitr *r)
{
    if (t->root < 0) return -1;
    int n = t->root;
    for (;;) {
        if (t->nodes[n].ch >= 0) return t->nodes[n].ch;
        uint32_t bit = br_read(r, 1);
        if (r->err) return -2;
        n = bit ? t->nodes[n].right : t->nodes[n].left;
        if (n < 0) return -1;
    }
}
/* huffman.rs:156-197 read */
static int huff_read(const htree *t, bitr *r, uint64_t max_bits)
{
    uint64_t remaining = max_bits - r->pos;
    int read_bits = remaining < 8 ? (int)remaining : 8;
    uint32_t cur = br_read(r, read_bits);
    if (r->err) return -2;
    hcode c = t->dec_table[cur & 255];
    if (c.len == 0) {
        r->pos -= (uint64_t)read_bits;
        return huff_read_slow(t, r);
    }
    r->pos = (uint64_t)((int64_t)r->pos - ((int64_t)read_bits - (int64_t)c.len));
    return c.symbol;
}

/* ---------------------------------------------------------------- src/lib.rs:96-158 test_entropy as a callable
 * encodes `n` coefficients as ONE run (like the test), returns the coded bytes; decode is the inverse */
PFVO_API size_t pfvo_entropy_roundtrip(const int16_t *data, size_t n, uint8_t *coded_out, size_t coded_cap, int16_t *decoded_out,
                                       uint8_t table_out[16])
{
    rle_vec seq = {0, 0, 0};
    rle_encode(&seq, data, n);
    int32_t table[16] = {0};
    update_table(table, seq.p, seq.n);
    htree tree;
    rle_create_huffman(&tree, table);
    memcpy(table_out, tree.table, 16);
    bytes out = {0, 0, 0};
    bitw w = {&out, 0, 0};
    for (size_t i = 0; i < seq.n; i++) {
        hcode z = tree.codes[seq.p[i].num_zeroes], b = tree.codes[seq.p[i].coeff_size];
        bw_write(&w, (int)z.len, z.val);
        bw_write(&w, (int)b.len, b.val);
        if (seq.p[i].coeff_size > 0) bw_write_signed(&w, seq.p[i].coeff_size, seq.p[i].coeff);
    }
    bw_align(&w);
    size_t len = out.n;
    if (len <= coded_cap) memcpy(coded_out, out.p, len);
    bitr r = {out.p, (uint64_t)out.n * 8, 0, 0};
    memset(decoded_out, 0, n * sizeof(int16_t));
    size_t idx = 0;
    while (idx < n) {
        int z = huff_read(&tree, &r, r.nbits);
        if (z < 0) break;
        idx += (size_t)z;
        int nb = huff_read(&tree, &r, r.nbits);
        if (nb < 0) break;
        if (nb > 0) { int32_t c = br_read_signed(&r, nb); if (idx < n) decoded_out[idx] = (int16_t)c; idx++; }
    }
    free(out.p); free(seq.p);
    return len;
}

/* ---------------------------------------------------------------- stream encoder (src/enc.rs) */
typedef struct {
    pfvo_encoder *hot;
    int width, height, framerate, quality, finished, total_blocks;
    int32_t q[4][64];
    bytes out;
    int16_t *coef; int8_t *mv; uint8_t *has;
} pfvo_stream_encoder;

/* enc.rs:190-219 write_header */
static void write_header(pfvo_stream_encoder *e)
{
    b_put(&e->out, "PFVIDEO\0", 8);                                               /* common.rs:1 */
    b_u32(&e->out, 211);                                                          /* common.rs:2 */
    b_u16(&e->out, (uint16_t)e->width); b_u16(&e->out, (uint16_t)e->height); b_u16(&e->out, (uint16_t)e->framerate);
    b_u16(&e->out, 4);
    for (int t = 0; t < 4; t++) for (int i = 0; i < 64; i++) b_u16(&e->out, (uint16_t)e->q[t][i]);
}

PFVO_API pfvo_stream_encoder *pfvo_stream_encoder_new(int width, int height, int framerate, int quality, int threads)
{
    pfvo_encoder *hot = pfvo_encoder_new(width, height, quality, threads);
    if (!hot) return NULL;
    pfvo_stream_encoder *e = (pfvo_stream_encoder *)calloc(1, sizeof *e);
    e->hot = hot; e->width = width; e->height = height; e->framerate = framerate; e->quality = quality;
    e->total_blocks = pfvo_encoder_total_blocks(hot);
    float px;
    pfvo_qtables(quality, e->q[0], e->q[1], e->q[2], e->q[3], &px);             /* header order: intra_l, intra_c, inter_l, inter_c */
    e->coef = (int16_t *)malloc((size_t)e->total_blocks * 512);
    e->mv = (int8_t *)malloc((size_t)e->total_blocks * 2);
    e->has = (uint8_t *)malloc((size_t)e->total_blocks);
    write_header(e);
    return e;
}

/* enc.rs:237-330 write_iframe_packet (payload only) */
static void serialize_iframe(const int16_t *coef, int total_blocks, bytes *payload)
{
    rle_vec *blocks = (rle_vec *)calloc((size_t)total_blocks, sizeof(rle_vec));
    int32_t table[16] = {0};
    for (int b = 0; b < total_blocks; b++) {
        rle_encode(&blocks[b], coef + (size_t)b * 256, 256);                      /* per macroblock (:246-255) */
        update_table(table, blocks[b].p, blocks[b].n);
    }
    htree tree;
    rle_create_huffman(&tree, table);
    bitw w = {payload, 0, 0};
    for (int i = 0; i < 16; i++) bw_write(&w, 8, tree.table[i]);                  /* :289-292 */
    bw_write(&w, 8, 0); bw_write(&w, 8, 1); bw_write(&w, 8, 1);                   /* :296-298 */
    for (int b = 0; b < total_blocks; b++) {
        for (size_t i = 0; i < blocks[b].n; i++) {
            rle_seq s = blocks[b].p[i];
            hcode z = tree.codes[s.num_zeroes], n = tree.codes[s.coeff_size];
            bw_write(&w, (int)z.len, z.val);
            bw_write(&w, (int)n.len, n.val);
            if (s.coeff_size > 0) bw_write_signed(&w, s.coeff_size, s.coeff);
        }
        free(blocks[b].p);
    }
    bw_align(&w);
    free(blocks);
}

/* enc.rs:332-481 write_pframe_packet (payload only) */
static void serialize_pframe(const int8_t *mv, const uint8_t *has, const int16_t *coef, int total_blocks, bytes *payload)
{
    rle_vec *blocks = (rle_vec *)calloc((size_t)total_blocks, sizeof(rle_vec));
    int32_t table[16] = {0};
    for (int b = 0; b < total_blocks; b++) {
        if (!has[b]) continue;                                                    /* None => nothing (:357-358) */
        rle_encode(&blocks[b], coef + (size_t)b * 256, 256);
        update_table(table, blocks[b].p, blocks[b].n);
    }
    htree tree;
    rle_create_huffman(&tree, table);
    bitw w = {payload, 0, 0};
    for (int i = 0; i < 16; i++) bw_write(&w, 8, tree.table[i]);
    bw_write(&w, 8, 2); bw_write(&w, 8, 3); bw_write(&w, 8, 3);                   /* :409-411 */
    for (int b = 0; b < total_blocks; b++) {                                      /* block headers (:414-451) */
        int has_mvec = mv[b * 2] != 0 || mv[b * 2 + 1] != 0;
        bw_write(&w, 1, (uint32_t)has_mvec);
        bw_write(&w, 1, has[b] ? 1u : 0u);
        if (has_mvec) { bw_write_signed(&w, 7, mv[b * 2]); bw_write_signed(&w, 7, mv[b * 2 + 1]); }
    }
    for (int b = 0; b < total_blocks; b++) {                                      /* :454-466 */
        for (size_t i = 0; i < blocks[b].n; i++) {
            rle_seq s = blocks[b].p[i];
            hcode z = tree.codes[s.num_zeroes], n = tree.codes[s.coeff_size];
            bw_write(&w, (int)z.len, z.val);
            bw_write(&w, (int)n.len, n.val);
            if (s.coeff_size > 0) bw_write_signed(&w, s.coeff_size, s.coeff);
        }
        free(blocks[b].p);
    }
    bw_align(&w);
    free(blocks);
}

static void write_packet(bytes *out, uint8_t type, const bytes *payload)
{
    b_u8(out, type);
    b_u32(out, payload ? (uint32_t)payload->n : 0u);
    if (payload && payload->n) b_put(out, payload->p, payload->n);
}

PFVO_API void pfvo_stream_encode_iframe(pfvo_stream_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    pfvo_encode_iframe(e->hot, y, u, v, e->coef);
    bytes payload = {0, 0, 0};
    serialize_iframe(e->coef, e->total_blocks, &payload);
    write_packet(&e->out, 1, &payload);                                           /* :323-327 */
    free(payload.p);
}
PFVO_API void pfvo_stream_encode_pframe(pfvo_stream_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    pfvo_encode_pframe(e->hot, y, u, v, e->mv, e->has, e->coef);
    bytes payload = {0, 0, 0};
    serialize_pframe(e->mv, e->has, e->coef, e->total_blocks, &payload);
    write_packet(&e->out, 2, &payload);                                           /* :474-478 */
    free(payload.p);
}
PFVO_API void pfvo_stream_encode_dropframe(pfvo_stream_encoder *e) { write_packet(&e->out, 1, NULL); }   /* :229-235 */
PFVO_API void pfvo_stream_finish(pfvo_stream_encoder *e)
{
    if (!e->finished) { e->finished = 1; write_packet(&e->out, 0, NULL); }        /* :182-188, 221-227 */
}
PFVO_API const uint8_t *pfvo_stream_bytes(const pfvo_stream_encoder *e, size_t *len) { *len = e->out.n; return e->out.p; }
PFVO_API void pfvo_stream_encoder_free(pfvo_stream_encoder *e)
{
    if (!e) return;
    pfvo_encoder_free(e->hot);
    free(e->out.p); free(e->coef); free(e->mv); free(e->has);
    free(e);
}
/* serialisers alone, for comparing the product's packet bytes on identical coefficient inputs */
PFVO_API size_t pfvo_serialize_iframe(const int16_t *coef, int total_blocks, uint8_t *out, size_t cap)
{
    bytes p = {0, 0, 0};
    serialize_iframe(coef, total_blocks, &p);
    size_t n = p.n;
    if (n <= cap) memcpy(out, p.p, n);
    free(p.p);
    return n;
}
PFVO_API size_t pfvo_serialize_pframe(const int8_t *mv, const uint8_t *has, const int16_t *coef, int total_blocks, uint8_t *out, size_t cap)
{
    bytes p = {0, 0, 0};
    serialize_pframe(mv, has, coef, total_blocks, &p);
    size_t n = p.n;
    if (n <= cap) memcpy(out, p.p, n);
    free(p.p);
    return n;
}

/* ---------------------------------------------------------------- stream decoder (src/dec.rs) */
typedef struct {
    const uint8_t *data; size_t len, pos, reset_pos;
    int width, height, framerate, n_qtables, eof, total_blocks;
    double delta_accum;
    pfvo_decoder *hot;
    int16_t *coef; int8_t *mv; uint8_t *has;
} pfvo_stream_decoder;

/* dec.rs:38-134 Decoder::new; err: -6 FormatError, -7 VersionError, -8 IOError */
PFVO_API pfvo_stream_decoder *pfvo_stream_decoder_new(const uint8_t *data, size_t len, int threads, int *err)
{
    *err = 0;
    if (len < 8) { *err = -8; return NULL; }
    if (memcmp(data, "PFVIDEO\0", 8) != 0) { *err = -6; return NULL; }
    if (len < 12) { *err = -8; return NULL; }
    uint32_t ver = (uint32_t)data[8] | ((uint32_t)data[9] << 8) | ((uint32_t)data[10] << 16) | ((uint32_t)data[11] << 24);
    if (ver != 211) { *err = -7; return NULL; }
    if (len < 20) { *err = -8; return NULL; }
    int w = data[12] | (data[13] << 8), h = data[14] | (data[15] << 8), fps = data[16] | (data[17] << 8), nq = data[18] | (data[19] << 8);
    if (len < 20 + (size_t)nq * 128) { *err = -8; return NULL; }
    int32_t *q = (int32_t *)malloc((size_t)(nq ? nq : 1) * 64 * sizeof(int32_t));
    for (int i = 0; i < nq * 64; i++) q[i] = data[20 + 2 * i] | (data[21 + 2 * i] << 8);
    pfvo_stream_decoder *d = (pfvo_stream_decoder *)calloc(1, sizeof *d);
    d->data = data; d->len = len; d->pos = d->reset_pos = 20 + (size_t)nq * 128;
    d->width = w; d->height = h; d->framerate = fps; d->n_qtables = nq;
    d->hot = pfvo_decoder_new(w, h, q, nq, threads);
    free(q);
    d->total_blocks = (pfvo_pad16(w) / 16) * (pfvo_pad16(h) / 16) + 2 * (pfvo_pad16(w / 2) / 16) * (pfvo_pad16(h / 2) / 16);
    d->coef = (int16_t *)malloc((size_t)d->total_blocks * 512);
    d->mv = (int8_t *)malloc((size_t)d->total_blocks * 2);
    d->has = (uint8_t *)malloc((size_t)d->total_blocks);
    return d;
}
PFVO_API void pfvo_stream_decoder_free(pfvo_stream_decoder *d)
{
    if (!d) return;
    pfvo_decoder_free(d->hot);
    free(d->coef); free(d->mv); free(d->has); free(d);
}
PFVO_API void pfvo_stream_decoder_info(const pfvo_stream_decoder *d, int *w, int *h, int *fps) { *w = d->width; *h = d->height; *fps = d->framerate; }
PFVO_API void pfvo_stream_decoder_reset(pfvo_stream_decoder *d) { d->eof = 0; d->pos = d->reset_pos; }   /* dec.rs:148-152 */

/* dec.rs:226-326 decode_iframe (bit parsing) */
static int parse_iframe(pfvo_stream_decoder *d, const uint8_t *payload, size_t n)
{
    bitr r = {payload, (uint64_t)n * 8, 0, 0};
    uint8_t table[16], qidx[3];
    for (int i = 0; i < 16; i++) table[i] = (uint8_t)br_read(&r, 8);
    htree tree;
    huff_from_table(&tree, table);
    for (int i = 0; i < 3; i++) qidx[i] = (uint8_t)br_read(&r, 8);
    if (r.err) return -8;
    for (int i = 0; i < 3; i++) if (qidx[i] >= d->n_qtables) return -6;
    size_t total = (size_t)d->total_blocks * 256, idx = 0;
    memset(d->coef, 0, total * sizeof(int16_t));
    while (idx < total) {                                                         /* :261-296: ONE run stream for the frame */
        int z = huff_read(&tree, &r, r.nbits);
        if (z < 0) return z == -2 ? -8 : -6;
        idx += (size_t)z;
        int nb = huff_read(&tree, &r, r.nbits);
        if (nb < 0) return nb == -2 ? -8 : -6;
        if (nb > 0) {
            int32_t c = br_read_signed(&r, nb);
            if (r.err) return -8;
            if (idx >= total) return -6;                                          /* the reference would index out of bounds */
            d->coef[idx++] = (int16_t)c;
        }
    }
    pfvo_decode_iframe(d->hot, d->coef, qidx);
    return 0;
}
/* dec.rs:328-448 decode_pframe */
static int parse_pframe(pfvo_stream_decoder *d, const uint8_t *payload, size_t n)
{
    bitr r = {payload, (uint64_t)n * 8, 0, 0};
    uint8_t table[16], qidx[3];
    for (int i = 0; i < 16; i++) table[i] = (uint8_t)br_read(&r, 8);
    htree tree;
    huff_from_table(&tree, table);
    for (int i = 0; i < 3; i++) qidx[i] = (uint8_t)br_read(&r, 8);
    if (r.err) return -8;
    for (int i = 0; i < 3; i++) if (qidx[i] >= d->n_qtables) return -6;
    for (int b = 0; b < d->total_blocks; b++) {                                   /* :361-372 */
        uint32_t has_mvec = br_read(&r, 1);
        d->has[b] = (uint8_t)br_read(&r, 1);
        d->mv[b * 2] = d->mv[b * 2 + 1] = 0;
        if (has_mvec) { d->mv[b * 2] = (int8_t)br_read_signed(&r, 7); d->mv[b * 2 + 1] = (int8_t)br_read_signed(&r, 7); }
        if (r.err) return -8;
    }
    memset(d->coef, 0, (size_t)d->total_blocks * 512);
    for (int b = 0; b < d->total_blocks; b++) {                                   /* :378-417 */
        if (!d->has[b]) continue;
        int16_t *blk = d->coef + (size_t)b * 256;
        size_t idx = 0;
        while (idx < 256) {
            int z = huff_read(&tree, &r, r.nbits);
            if (z < 0) return z == -2 ? -8 : -6;
            idx += (size_t)z;
            int nb = huff_read(&tree, &r, r.nbits);
            if (nb < 0) return nb == -2 ? -8 : -6;
            if (nb > 0) {
                int32_t c = br_read_signed(&r, nb);
                if (r.err) return -8;
                if (idx >= 256) return -6;
                blk[idx++] = (int16_t)c;
            }
        }
    }
    return pfvo_decode_pframe(d->hot, d->mv, d->has, d->coef, qidx) ? -4 : 0;
}

static void crop_retframe(const pfvo_stream_decoder *d, uint8_t *frame_out)       /* dec.rs:195-197, 209-211 */
{
    size_t o = 0;
    for (int p = 0; p < 3; p++) {
        int pw, ph;
        const uint8_t *src = pfvo_decoder_plane(d->hot, p, &pw, &ph);
        int w = p == 0 ? d->width : d->width / 2, h = p == 0 ? d->height : d->height / 2;
        for (int r = 0; r < h; r++) { memcpy(frame_out + o, src + (size_t)r * pw, (size_t)w); o += (size_t)w; }
    }
}

/* dec.rs:169-224 advance_frame.  Returns 1 (Ok(true)), 0 (Ok(false): EOF), negative error.
 * *got_frame = 1 when onvideo would have been called (frame_out then holds the retframe, Y|U|V). */
PFVO_API int pfvo_stream_advance_frame(pfvo_stream_decoder *d, uint8_t *frame_out, int *got_frame)
{
    *got_frame = 0;
    if (d->eof) return 0;
    for (;;) {
        if (d->pos + 5 > d->len) return -8;
        uint8_t type = d->data[d->pos];
        uint32_t plen = (uint32_t)d->data[d->pos + 1] | ((uint32_t)d->data[d->pos + 2] << 8) | ((uint32_t)d->data[d->pos + 3] << 16) |
                        ((uint32_t)d->data[d->pos + 4] << 24);
        d->pos += 5;
        if (type == 0) { d->eof = 1; return 0; }
        if (type == 1 || type == 2) {
            if (type == 1 && plen == 0) break;                                    /* drop frame (:190) */
            if (d->pos + plen > d->len) return -8;
            int rc = type == 1 ? parse_iframe(d, d->data + d->pos, plen) : parse_pframe(d, d->data + d->pos, plen);
            d->pos += plen;
            if (rc) return rc;
            crop_retframe(d, frame_out);
            *got_frame = 1;
            break;
        }
        if (d->pos + plen > d->len) return -8;                                    /* unknown packet: skip (:216-219) */
        d->pos += plen;
    }
    return 1;
}
