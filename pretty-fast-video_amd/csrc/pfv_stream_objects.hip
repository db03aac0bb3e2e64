// pfv_stream_objects.hip -- stream-level objects, first half: pfv_encoder (enc::Encoder<W>), the staging types the three decoder objects share, the host half of the decoders' device entropy stage.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
// ================================================================== stream-level session objects
// enc::Encoder<W> (src/enc.rs:12-188) with W = an in-memory byte vector (the reference's tests use
// Cursor<Vec<u8>>, src/lib.rs:319-321), dec::Decoder<R> (src/dec.rs:15-224) with R = a caller-owned byte slice.
// Page-locked host staging (hipHostMalloc): PCIe copies from / to these run at link rate without the runtime's
// bounce through its own pinned chunks; where page-locking is refused the buffer is ordinary memory.
template <class T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    bool pinned = false;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release()
    {
        if (p && pinned) (void)hipHostFree(p);
        else if (p) free(p);
        p = nullptr; n = 0;
    }
    bool resize(size_t count)
    {
        if (count <= n) return true;
        release();
        if (hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault) == hipSuccess) {
            pinned = true;
        } else {   // locked-memory limits: pageable memory still works, the copies just bounce through the runtime
            (void)hipGetLastError();
            p = (T *)malloc(count * sizeof(T));
            pinned = false;
            if (!p) return false;
        }
        n = count;
        return true;
    }
    T *data() { return p; }
    size_t size() const { return n; }
    void swap(PinnedBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(pinned, o.pinned); }
};

struct pfv_encoder {
    pfv_ctx *ctx = nullptr;
    pfv_enc_session *hot = nullptr;
    int width = 0, height = 0, framerate = 0, total_blocks = 0;
    bool finished = false;
    bool device_entropy = true;            // payloads built by the k_ent_* kernels instead of serialize_*frame on the host
    const uint8_t *plane[3] = {nullptr, nullptr, nullptr};   // device path: the caller's planes of the frame being encoded
    bool poisoned = false;                 // a frame failed after prev_frame had moved on: the next frame must be an i-frame
    std::vector<uint8_t> out;              // the writer: bytes produced and not yet handed over (pfv_encoder_drain)
    std::vector<uint8_t> drained;          // what the last pfv_encoder_drain handed over
    PinnedBuf<uint8_t> frame;              // packed Y|U|V staging
    PinnedBuf<int16_t> coef;               // host entropy path only
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    PinnedBuf<uint8_t> payload;            // device entropy path: packet payload landing zone
};

// One step of Decoder::advance_frame's packet loop (src/dec.rs:169-224), found by the header scanner.  FRAME events are
// parsed (bits -> coefficients / block headers, dec.rs:226-296, 328-417) ahead of their turn by worker threads: packets
// are independent bit streams, only the device decode behind them is sequential.
// ------------------------------------------------------------------ the decoders' entropy stage on the device: host half
// What the host reads of a packet for the k_entd_* kernels (pfv_entdec_kernels.hip): its first 19 bytes -- the table (-> the tree's codes) and
// the q indices.  A p-frame's block headers are read on the device since round 5 (k_hdr_*: motion vectors, has_coeff, the first bit of the run
// streams), the list of coded macroblocks is made there from the has_coeff bytes (k_entd_coded).
// The payload is copied to `bytes_dst` (page-locked staging, >= plen + 16 bytes).  The caller has set k.byte_off / k.frame_off.
// header workgroups (k_hdr_*) of a p-frame packet: 2 048 bits each, as many as its headers can take (16 bits per macroblock) or its payload has
static inline uint32_t entd_hdr_wgs(size_t tb, size_t plen)
{
    const size_t bits = plen * 8 > kHdrBit0 ? plen * 8 - kHdrBit0 : 0;
    return (uint32_t)((std::min(bits, tb * 16) + kHdrWgBits - 1) / kHdrWgBits);
}
struct EntdPrep {
    int rc = 0;                  // a status the host parser would have returned before it read any run (header, q index, truncated block headers)
    bool host_parse = false;     // the host parser has to read this packet (degenerate code table, 512 MiB or more, no bits behind the headers)
    uint8_t qidx[3] = {0, 0, 0};
};
static EntdPrep entd_prepare(const uint8_t *payload, uint32_t plen, int type, size_t tb, int n_qtables, uint32_t sub_bits, EdPacket &k, uint8_t *bytes_dst)
{
    EntdPrep p;
    k.total_bits = k.bit0 = k.total_coefs = k.n_sub = k.sub_first = k.grp_first = k.list_cap = 0;
    k.org = k.first_sub = k.hdr_first = k.hdr_wgs = 0;
    k.sub_bits = sub_bits;
    k.pframe = type == 2 ? 1u : 0u;
    k.total_blocks = (uint32_t)tb;
    memset(k.code_val, 0, sizeof k.code_val);
    memset(k.code_len, 0, sizeof k.code_len);
    BitSource r(payload, plen);
    PacketHead h;
    p.rc = parse_head(r, h, n_qtables);
    if (p.rc) return p;
    memcpy(p.qidx, h.qidx, 3);
    int n_syms = 0;
    for (uint8_t t : h.table) n_syms += t != 0;
    const uint64_t bits = (uint64_t)plen * 8, bit0 = r.position();      // behind the table and the q indices: bit 152
    // zero-length codes / no bits left / 64 MiB and more: a run costs two bits or more and covers at most 16 coefficients, so below 2^29 bits
    // the kernels' counters (coefficients and values per packet, 32 bits each, summed side by side in one 64-bit word) cannot overflow
    if (n_syms < 2 || bits >= (1ull << 29) || bit0 >= bits) { p.host_parse = true; return p; }
    HuffmanTree tree(h.table);
    for (int s = 0; s < 16; s++) {
        k.code_val[s] = (uint16_t)tree.code((uint8_t)s).val;
        k.code_len[s] = (uint8_t)tree.code((uint8_t)s).len;
    }
    k.total_bits = (uint32_t)bits;
    k.bit0 = k.org = (uint32_t)bit0;
    k.n_sub = (uint32_t)((bits - bit0 + sub_bits - 1) / sub_bits);
    if (type == 2) {
        // the block headers (src/dec.rs:351-372) are read on the device (k_hdr_*): where the run streams start, how many macroblocks are coded
        // and what the list can need is written into the descriptor there; the subsequences are counted from bit 152
        k.hdr_wgs = entd_hdr_wgs(tb, plen);
    } else {
        k.total_coefs = (uint32_t)(tb * 256);
        k.list_cap = (uint32_t)std::min<uint64_t>(tb * 256, (bits - bit0) / 3 + 1);   // <= entd_pool_cap(tb, plen): the room the caller set aside
    }
    memcpy(bytes_dst, payload, plen);
    memset(bytes_dst + plen, 0, 16);
    return p;
}
// Entries a packet's coefficient list can need, known before any of it is read: a value costs three bits or more (two tree codes of a bit or
// more -- tables of fewer than two symbols go to the host parser -- and coeff_size >= 1 value bits), and there are no more values than
// coefficients.  Rounded up to whole 16-byte lines so that the lists of a pool start aligned.
static inline size_t entd_pool_cap(size_t tb, size_t plen) { return (std::min(tb * 256, plen * 8 / 3 + 1) + 3) & ~(size_t)3; }

// Device side of the coefficient lists of `frames` frames (pfv_device.h: CoefLists): a pool of entries the frames' lists are cut from, the
// table of list pointers the decode kernels index by slot, the frames' counts.  A list that does not fit its place in the pool -- only a
// packet the HOST parser read can need more than entd_pool_cap (a one-symbol table: values of one or two bits) -- gets a buffer of its own
// for the life of the batch (spill).
struct ListPool {
    uint32_t *ent = nullptr; size_t ent_cap = 0;       // entries
    uint32_t **ptr_dev = nullptr;                      // [frames]
    uint32_t *counts_dev = nullptr;                    // [frames][tb + 1]
    size_t frames = 0, tb = 0;
    PinnedBuf<uint32_t *> ptr_host;
    std::vector<uint32_t *> spill;
    long spilled = 0;                                  // lists that got a buffer of their own so far
    int create(pfv_ctx *ctx, size_t n_frames, size_t total_blocks, size_t entries)
    {
        frames = n_frames; tb = total_blocks;
        HIP_TRY(ctx, hipMalloc((void **)&ptr_dev, n_frames * sizeof(uint32_t *)));
        HIP_TRY(ctx, hipMalloc((void **)&counts_dev, n_frames * (total_blocks + 1) * sizeof(uint32_t)));
        if (entries) { HIP_TRY(ctx, hipMalloc((void **)&ent, entries * sizeof(uint32_t))); ent_cap = entries; }
        if (!ptr_host.resize(n_frames)) return fail(ctx, PFV_ERR_NOMEM, "pinned list-pointer staging");
        for (size_t f = 0; f < n_frames; f++) ptr_host.data()[f] = nullptr;
        return PFV_OK;
    }
    // room for `entries` in the pool; the caller has made sure nothing on the device still uses it
    int room(pfv_ctx *ctx, size_t entries)
    {
        if (entries <= ent_cap) return PFV_OK;
        if (ent) { (void)hipFree(ent); ent = nullptr; ent_cap = 0; }
        entries += entries / 4;
        HIP_TRY(ctx, hipMalloc((void **)&ent, entries * sizeof(uint32_t)));
        ent_cap = entries;
        return PFV_OK;
    }
    void drop_spill()
    {
        for (uint32_t *p : spill) (void)hipFree(p);
        spill.clear();
    }
    void destroy()
    {
        drop_spill();
        for (void *p : {(void *)ent, (void *)ptr_dev, (void *)counts_dev})
            if (p) (void)hipFree(p);
        ent = nullptr; ptr_dev = nullptr; counts_dev = nullptr; ent_cap = 0;
    }
    DecCoefs coefs(size_t first_frame = 0) const { return DecCoefs(ptr_dev + first_frame, counts_dev + first_frame * (tb + 1)); }
};

// A packet through the HOST parser into list form, for a decoder whose coefficients travel as lists: entries and counts into page-locked
// staging (`ent` with room for `cap` entries, `counts` [tb + 1]).  kSinkFull: more than `cap` entries (parse again with room for tb x 256).
static int parse_to_lists(const uint8_t *payload, size_t plen, int type, size_t tb, int n_qtables, int8_t *mv, uint8_t *has, uint32_t *ent, size_t cap, uint32_t *counts,
                          size_t *n_out, uint8_t qidx[3])
{
    ListSink sink{ent, cap, counts, tb};
    const int rc = type == 2 ? parse_pframe_to(payload, plen, (int)tb, n_qtables, mv, has, sink, qidx) : parse_iframe_to(payload, plen, (int)tb, n_qtables, sink, qidx);
    sink.finish();
    *n_out = sink.n;
    return rc;
}
// ... and onto the device, in frame `f`'s place of the pool (or a buffer of its own when it is longer than the place: `place_cap` entries),
// on `stream`; the staging is free again when the stream has passed this point
static int upload_lists(pfv_ctx *ctx, ListPool &lp, size_t f, size_t place_cap, const uint32_t *ent, size_t n, const uint32_t *counts, hipStream_t stream)
{
    uint32_t *dst = lp.ptr_host.data()[f];
    if (n > place_cap || !dst) {
        HIP_TRY(ctx, hipMalloc((void **)&dst, std::max<size_t>(n, 1) * sizeof(uint32_t)));
        lp.spill.push_back(dst);
        lp.spilled++;
        lp.ptr_host.data()[f] = dst;
        HIP_TRY(ctx, hipMemcpyAsync(lp.ptr_dev + f, lp.ptr_host.data() + f, sizeof(uint32_t *), hipMemcpyHostToDevice, stream));
    }
    if (n) HIP_TRY(ctx, hipMemcpyAsync(dst, ent, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(lp.counts_dev + f * (lp.tb + 1), counts, (lp.tb + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    return PFV_OK;
}

// the launches of one window: np packets from b.packet0 on, ng workgroups from b.group0 on (b.groups already points at the first of them)
static void entd_launch(hipStream_t stream, const EdBufs &b, unsigned np, unsigned ng, unsigned max_hdr_wgs, int launches, int inner)
{
    if (max_hdr_wgs) {   // the p-frames' block headers first: they complete the packet descriptors the kernels below read
        hipLaunchKernelGGL(k_hdr_map, dim3(max_hdr_wgs, np), dim3(kEdThreads), 0, stream, b);
        hipLaunchKernelGGL(k_hdr_scan, dim3(np), dim3(kEdThreads), 0, stream, b);
        hipLaunchKernelGGL(k_hdr_emit, dim3(max_hdr_wgs, np), dim3(kEdThreads), 0, stream, b);
    }
    hipLaunchKernelGGL(k_entd_coded, dim3(np), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_sync, dim3(ng), dim3(kEdThreads), 0, stream, b, inner);                      // every subsequence, settled inside the workgroups
    for (int round = 1; round < launches; round++)                                                           // the seams between them (a second pass finds nothing, as a rule)
        hipLaunchKernelGGL(k_entd_fix, dim3((ng + kEdFixThreads - 1) / kEdFixThreads), dim3(kEdFixThreads), 0, stream, b, (uint32_t)ng);
    hipLaunchKernelGGL(k_entd_verify, dim3(ng), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_prefix, dim3(np), dim3(kEdThreads), 0, stream, b);
    hipLaunchKernelGGL(k_entd_emit, dim3(ng), dim3(kEdThreads), 0, stream, b);
}

struct DecEvent {
    enum Kind { FRAME, DROP, END, ERROR } kind = END;
    enum State { FREE, QUEUED, RUNNING, DONE } state = FREE;
    int rc = 0;                          // ERROR: the status to return; FRAME: parse result
    const char *msg = "";
    uint8_t type = 0;                    // FRAME: 1 = i-frame, 2 = p-frame
    size_t pos_after = 0;                // stream position once this event has been consumed
    const uint8_t *payload = nullptr;
    uint32_t plen = 0;
    uint8_t qidx[3] = {0, 0, 0};
    PinnedBuf<int16_t> coef;             // dense form: only when the sparse list overflowed
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    PinnedBuf<uint32_t> idx;             // sparse form: non-zero coefficients as (flat index, value)
    PinnedBuf<int16_t> val;
    size_t n_sparse = 0;
    bool dense = false;
    // device-entropy form (PFV_OPT_ENTROPY_DECODE): what entd_prepare left for the k_entd_* kernels instead of a parsed packet
    bool dev_form = false, host_parse = false;
    PinnedBuf<uint8_t> bytes;            // the payload (+ 16)
    PinnedBuf<EdPacket> pk;              // 1
    PinnedBuf<uint2> groups;             // workgroups of the packet
};

// switches, shape and counters of the device entropy stage in pfv_decoder / pfv_batch_decoder (the buffers: DecWindow)
struct DecEntd {
    std::atomic<bool> on{false};         // read by the parser threads; cleared by the caller's thread when the window sets cannot be made (AUTO: the host parser takes over)
    bool force = false;                  // force: every packet (PFV_ENTROPY_DECODE_DEVICE); otherwise payloads of kDecEntdMinBytes and more
    bool ready = false;                  // the window stream and the window sets exist: made by the first packet / step that takes the device form
    //                                      (a decoder of small packets never needs them), entd_windows_make
    uint32_t sub_bits = kEdSubBits;
    int launches = 3, inner = kEdInner;
    long packets_dev = 0, packets_host = 0;
};
constexpr uint32_t kDecEntdMinBytes = 64 * 1024;   // below this the launches cost more than the host parser needs for the packet

// device side of one packet's window in pfv_decoder.  Two alternate: the window of the NEXT packet (uploads, k_entd_*, status) runs on a
// second stream under the decode launch and the frame download of the current one.
struct DecWindow {
    uint8_t *bytes_dev = nullptr; size_t bytes_cap = 0;
    uint2 *groups_dev = nullptr; size_t groups_cap = 0;
    uint32_t *sub_dev = nullptr; size_t sub_cap = 0;
    EdPacket *pk_dev = nullptr;
    uint32_t *status_dev = nullptr, *coded_dev = nullptr;
    unsigned long long *wgsum_dev = nullptr; size_t wgsum_cap = 0;
    uint32_t *hdr_maps_dev = nullptr; size_t hdr_maps_cap = 0;      // k_hdr_*: [header workgroup][8]
    uint4 *hdr_start_dev = nullptr; size_t hdr_start_cap = 0;       // [header workgroup]
    ListPool lists;                      // the window's coefficients: one list per packet (pfv_device.h: CoefLists)
    std::vector<size_t> list_room;       // per packet: the size of its list's place in the pool
    int8_t *mv_dev = nullptr;
    uint8_t *has_dev = nullptr;
    PinnedBuf<uint32_t> status_host;
    hipEvent_t done = nullptr;
    DecEvent *owner = nullptr;           // the packet whose window is enqueued / was decoded from this set
    void destroy()
    {
        for (void *p : {(void *)bytes_dev, (void *)pk_dev, (void *)status_dev, (void *)coded_dev, (void *)groups_dev, (void *)sub_dev, (void *)wgsum_dev, (void *)mv_dev, (void *)has_dev,
                        (void *)hdr_maps_dev, (void *)hdr_start_dev})
            if (p) (void)hipFree(p);
        lists.destroy();
        if (done) (void)hipEventDestroy(done);
    }
};

// The window stream and the fixed-size part of every window set, for S packets per window: on the caller's thread, when the first packet (step)
// takes the device form.
template <size_t N>
static int entd_windows_make(pfv_ctx *ctx, DecEntd &v, DecWindow (&win)[N], hipStream_t *stream, size_t S, size_t tb)
{
    if (v.ready) return PFV_OK;
    hipError_t he = *stream ? hipSuccess : hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    bool host_ok = true;
    for (DecWindow &w : win) {
        if (he == hipSuccess && !w.pk_dev) he = hipMalloc((void **)&w.pk_dev, S * sizeof(EdPacket));
        if (he == hipSuccess && !w.status_dev) he = hipMalloc((void **)&w.status_dev, S * sizeof(uint32_t));
        if (he == hipSuccess && !w.coded_dev) he = hipMalloc((void **)&w.coded_dev, S * tb * sizeof(uint32_t));
        if (he == hipSuccess && !w.lists.ptr_dev && w.lists.create(ctx, S, tb, 0) != PFV_OK) he = hipErrorOutOfMemory;
        if (he == hipSuccess && !w.mv_dev) he = hipMalloc((void **)&w.mv_dev, S * tb * 2);
        if (he == hipSuccess && !w.has_dev) he = hipMalloc((void **)&w.has_dev, S * tb);
        if (he == hipSuccess && !w.done) he = hipEventCreateWithFlags(&w.done, hipEventDisableTiming);
        host_ok = host_ok && w.status_host.resize(S);
    }
    if (he != hipSuccess) return hip_fail(ctx, he, "device entropy stage: window sets");
    if (!host_ok) return fail(ctx, PFV_ERR_NOMEM, "device entropy stage: pinned status words");
    v.ready = true;
    return PFV_OK;
}
// host staging of one packet the host parser reads into list form (a decoder whose coefficients travel as lists)
struct ListStage {
    PinnedBuf<uint32_t> ent, counts;
    size_t n = 0;
    // kSinkFull cannot come back: a list of the place's size is tried first, then one with room for every coefficient
    int parse(const uint8_t *payload, size_t plen, int type, size_t tb, int n_qtables, int8_t *mv, uint8_t *has, size_t place_cap, uint8_t qidx[3])
    {
        if (!ent.resize(std::max<size_t>(place_cap, 4)) || !counts.resize(tb + 1)) return PFV_ERR_NOMEM;
        int rc = parse_to_lists(payload, plen, type, tb, n_qtables, mv, has, ent.data(), place_cap, counts.data(), &n, qidx);
        if (rc != kSinkFull) return rc;
        if (!ent.resize(tb * 256)) return PFV_ERR_NOMEM;
        return parse_to_lists(payload, plen, type, tb, n_qtables, mv, has, ent.data(), tb * 256, counts.data(), &n, qidx);
    }
};

constexpr int kDecWindows = 4;           // pfv_decoder: windows in flight -- the packet being decoded and up to three behind it
struct pfv_decoder {
    DecEntd entd;                        // switches, shape and counters of the device entropy stage (its buffers: win[])
    DecWindow win[kDecWindows];
    ListStage hp;                        // a packet the device stage left to the host parser
    hipStream_t win_stream = nullptr;
    pfv_ctx *ctx = nullptr;
    pfv_dec_session *hot = nullptr;
    const uint8_t *data = nullptr;
    size_t len = 0, pos = 0, reset_pos = 0;
    int width = 0, height = 0, framerate = 0, n_qtables = 0, total_blocks = 0;
    bool eof = false;
    double delta_accum = 0.0;
    PinnedBuf<uint8_t> retframe;           // Y|U|V, unpadded (src/dec.rs:22)
    uint8_t *frame_dev = nullptr;          // pfv_decoder_set_output_device: the retframe in device memory instead
    // look-ahead: ring of events in stream order, [head, head + count)
    std::vector<std::unique_ptr<DecEvent>> ring;
    size_t head = 0, count = 0;
    size_t scan_pos = 0;
    bool scan_stop = false;                // an END / ERROR event is pending: nothing is scanned past it
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    bool quit = false;
};

static void put_u16(std::vector<uint8_t> &o, unsigned v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
static void put_u32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_packet(std::vector<uint8_t> &o, uint8_t type, const std::vector<uint8_t> *payload)
{
    o.push_back(type);
    put_u32(o, payload ? (uint32_t)payload->size() : 0u);
    if (payload) o.insert(o.end(), payload->begin(), payload->end());
}

extern "C" {

// Encoder::new (src/enc.rs:37-73): q-tables from quality, prev_frame = new_padded, write_header (:190-219)
PFV_API int pfv_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, pfv_encoder **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_encoder_create: bad argument");
    *out = nullptr;
    if (framerate < 0 || framerate > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "framerate must fit u16 (src/enc.rs:197)");
    pfv_enc_session *hot = nullptr;
    int rc = pfv_enc_session_create(ctx, width, height, quality, 1, &hot);
    if (rc) return rc;
    pfv_encoder *e = new pfv_encoder();
    e->ctx = ctx; e->hot = hot; e->width = width; e->height = height; e->framerate = framerate;
    e->total_blocks = pfv_total_blocks(width, height);
    if (!e->frame.resize(pfv_frame_bytes(width, height))) {
        pfv_encoder_destroy(e);
        return fail(ctx, PFV_ERR_NOMEM, "pfv_encoder_create: pinned staging");
    }
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], nullptr);
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};      // common.rs:1
    e->out.insert(e->out.end(), magic, magic + 8);
    put_u32(e->out, 211);                                                      // common.rs:2
    put_u16(e->out, (unsigned)width); put_u16(e->out, (unsigned)height); put_u16(e->out, (unsigned)framerate);
    put_u16(e->out, 4);
    for (int t = 0; t < 4; t++)                                                // intra_l, intra_c, inter_l, inter_c
        for (int i = 0; i < 64; i++) put_u16(e->out, (unsigned)q[t][i]);
    *out = e;
    return PFV_OK;
}

static int pack_frame(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!y || !u || !v) return fail(e->ctx, PFV_ERR_BAD_ARG, "null plane");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:80)");
    if (e->device_entropy) {   // the planes go up from where they lie (encode_on_device): no packing copy -- it was half of a 4K frame's time
        e->plane[0] = y; e->plane[1] = u; e->plane[2] = v;
        return PFV_OK;
    }
    size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
    memcpy(e->frame.data(), y, ny);
    memcpy(e->frame.data() + ny, u, nc);
    memcpy(e->frame.data() + ny + nc, v, nc);
    return PFV_OK;
}

static int host_entropy_staging(pfv_encoder *e)
{
    if (e->coef.resize((size_t)e->total_blocks * 256) && e->mv.resize((size_t)e->total_blocks * 2) && e->has.resize((size_t)e->total_blocks))
        return PFV_OK;
    return fail(e->ctx, PFV_ERR_NOMEM, "pinned staging for the host entropy path");
}

// One frame through the device entropy stage: planes up, kernels, payload size then payload bytes down.
static int encode_on_device(pfv_encoder *e, bool pframe)
{
    pfv_enc_session *s = e->hot;
    pfv_ctx *ctx = e->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = enc_staging(s);
    if (!rc) rc = pfv_enc_entropy_enable(s, 0);
    if (rc) return rc;
    {   // the caller's planes are read until the first synchronisation below (pfv_enc_payload_sizes); every exit before it synchronises too
        const size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
        const bool packed = e->plane[1] == e->plane[0] + ny && e->plane[2] == e->plane[1] + nc;   // a packed frame: one copy
        hipError_t he = hipMemcpyAsync(s->st_frames, e->plane[0], packed ? ny + 2 * nc : ny, hipMemcpyHostToDevice, ctx->stream);
        if (!packed && he == hipSuccess) he = hipMemcpyAsync(s->st_frames + ny, e->plane[1], nc, hipMemcpyHostToDevice, ctx->stream);
        if (!packed && he == hipSuccess) he = hipMemcpyAsync(s->st_frames + ny + nc, e->plane[2], nc, hipMemcpyHostToDevice, ctx->stream);
        if (he != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return hip_fail(ctx, he, "plane upload"); }
    }
    rc = pframe ? pfv_enc_pframe_dev(s, s->st_frames, s->st_mv, s->st_has, s->st_coef) : pfv_enc_iframe_dev(s, s->st_frames, s->st_coef);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    // from here on prev_frame has moved to this frame: a failure leaves the encoder's reference ahead of the stream
    e->poisoned = true;
    rc = pframe ? pfv_enc_pack_pframe_dev(s, s->st_mv, s->st_has, s->st_coef) : pfv_enc_pack_iframe_dev(s, s->st_coef);
    uint32_t nbytes = 0;
    if (!rc) rc = pfv_enc_payload_sizes(s, &nbytes);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if (!e->payload.resize(std::max<size_t>(nbytes, 1 << 20))) return fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
    if ((rc = pfv_enc_payload_fetch(s, 0, e->payload.data(), nbytes))) return rc;
    e->poisoned = false;
    e->out.push_back(pframe ? 2 : 1);
    put_u32(e->out, nbytes);
    e->out.insert(e->out.end(), e->payload.data(), e->payload.data() + nbytes);
    return PFV_OK;
}

// 1 (default): RLE + Huffman + bit packing on the device; 0: on the host (serialize_iframe / serialize_pframe).  The
// bytes written are the same either way.
PFV_API int pfv_encoder_set_device_entropy(pfv_encoder *e, int on)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    e->device_entropy = on != 0;
    return PFV_OK;
}

// Encoder::encode_iframe (src/enc.rs:75-123)
PFV_API int pfv_encoder_encode_iframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    int rc = pack_frame(e, y, u, v);
    if (rc) return rc;
    if (e->device_entropy) return encode_on_device(e, false);      // an i-frame replaces prev_frame entirely: clears a poisoned state
    if ((rc = host_entropy_staging(e))) return rc;
    if ((rc = pfv_enc_iframe(e->hot, e->frame.data(), e->coef.data()))) return rc;
    e->poisoned = true;
    std::vector<uint8_t> payload;
    if (!serialize_iframe(payload, e->coef.data(), e->total_blocks))
        return fail(e->ctx, PFV_ERR_FORMAT, "coefficient needs more than 15 size bits (src/rle.rs:44)");
    put_packet(e->out, 1, &payload);
    e->poisoned = false;
    return PFV_OK;
}
// Encoder::encode_pframe (src/enc.rs:125-173)
PFV_API int pfv_encoder_encode_pframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    int rc = pack_frame(e, y, u, v);
    if (rc) return rc;
    // a previous frame failed after the encoder's reference had advanced but before its packet was written: a p-frame
    // now would predict from a frame the decoder never saw (the reference panics in that situation and the Encoder is gone)
    if (e->poisoned) return fail(e->ctx, PFV_ERR_STATE, "the previous frame failed after prev_frame had advanced: encode an i-frame next");
    if (e->device_entropy) return encode_on_device(e, true);
    if ((rc = host_entropy_staging(e))) return rc;
    if ((rc = pfv_enc_pframe(e->hot, e->frame.data(), e->mv.data(), e->has.data(), e->coef.data()))) return rc;
    e->poisoned = true;
    std::vector<uint8_t> payload;
    if (!serialize_pframe(payload, e->mv.data(), e->has.data(), e->coef.data(), e->total_blocks))
        return fail(e->ctx, PFV_ERR_FORMAT, "coefficient needs more than 15 size bits (src/rle.rs:44)");
    put_packet(e->out, 2, &payload);
    e->poisoned = false;
    return PFV_OK;
}
// Encoder::encode_dropframe (src/enc.rs:175-180): an i-frame packet with an empty payload
PFV_API int pfv_encoder_encode_dropframe(pfv_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:176)");
    put_packet(e->out, 1, nullptr);
    return PFV_OK;
}
// Encoder::finish (src/enc.rs:182-188): EOF packet
PFV_API int pfv_encoder_finish(pfv_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:183)");
    e->finished = true;
    put_packet(e->out, 0, nullptr);
    return PFV_OK;
}
PFV_API int pfv_encoder_bytes(pfv_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_encoder_bytes: bad argument");
    *data = e->out.data();
    *len = e->out.size();
    return PFV_OK;
}
// The reference streams every packet to its writer and keeps nothing (src/enc.rs:190-235); so does this: the bytes produced
// since the last drain are handed over and forgotten, only the current packet is ever resident.
PFV_API int pfv_encoder_drain(pfv_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_encoder_drain: bad argument");
    e->drained.swap(e->out);
    e->out.clear();
    *data = e->drained.data();
    *len = e->drained.size();
    return PFV_OK;
}
// Drop for Encoder (src/enc.rs:28-34): finishes the stream if the caller did not
PFV_API void pfv_encoder_destroy(pfv_encoder *e)
{
    if (!e) return;
    pfv_enc_session_destroy(e->hot);
    delete e;
}
