// pfv_batch_objects.hip -- pfv_batch_encoder / pfv_batch_decoder: n streams of one geometry stepped together.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
// ------------------------------------------------------------------ batch encoder: n streams, pipelined
// n independent streams of one geometry encoded together (the reference runs one Encoder per stream, src/enc.rs:12-26):
// per frame step ONE upload, one launch per stage for all streams, one download of all payloads.  The upload of step t runs
// on its own copy stream while the host collects step t-1 (payload download, packet assembly, writers) and before the
// kernels of step t are enqueued, so PCIe, the kernels and the host work of neighbouring steps overlap:
//     encode(t):  [copy stream] frames(t) -> HBM      [host] finish step t-1: payloads -> writers
//                 [main stream] wait upload(t); k_enc_*; k_ent_*       (returns without waiting for them)
// Every writer receives exactly the bytes an Encoder of its own would have written (packets arrive one step late; finish
// flushes).
struct pfv_batch_encoder {
    pfv_ctx *ctx = nullptr;
    pfv_enc_session *hot = nullptr;
    int n = 0, width = 0, height = 0;
    size_t frame_bytes = 0, total_blocks = 0;
    pfv_write_cb write = nullptr;
    void *user = nullptr;
    std::vector<std::vector<uint8_t>> kept;   // write == NULL: per-stream bytes until pfv_batch_encoder_take
    std::vector<std::vector<uint8_t>> taken;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    uint8_t *in_host[2] = {nullptr, nullptr};   // page-locked [n][frame_bytes], filled by the caller
    uint8_t *in_dev[2] = {nullptr, nullptr};
    int16_t *coef = nullptr;
    int8_t *mv = nullptr;
    uint8_t *has = nullptr;
    PinnedBuf<uint8_t> payloads;
    std::vector<uint32_t> sizes;
    std::vector<uint64_t> offsets;
    std::vector<uint8_t> packet;
    long step = 0;
    int pending = -1;          // packet type of the step whose kernels are in flight, -1: none
    bool finished = false, poisoned = false;
};

static void be_emit(pfv_batch_encoder *b, int stream, const uint8_t *data, size_t len)
{
    if (b->write) b->write(b->user, stream, data, len);
    else b->kept[(size_t)stream].insert(b->kept[(size_t)stream].end(), data, data + len);
}
// the step in flight: wait for it, fetch every payload with one copy, hand the packets to the writers
static int be_collect(pfv_batch_encoder *b)
{
    if (b->pending < 0) return PFV_OK;
    const int type = b->pending;
    b->pending = -1;
    int rc = pfv_enc_payloads_fetch(b->hot, b->payloads.data(), b->payloads.size(), b->sizes.data(), b->offsets.data());
    if (rc == PFV_ERR_NOMEM) {   // very dense content: retry with the worst-case landing zone
        const size_t worst = (size_t)b->n * ((pfv_payload_worst_case(b->width, b->height) + 15) & ~(size_t)15);
        if (b->payloads.size() < worst && b->payloads.resize(worst))
            rc = pfv_enc_payloads_fetch(b->hot, b->payloads.data(), b->payloads.size(), b->sizes.data(), b->offsets.data());
    }
    if (rc) { b->poisoned = true; return rc; }
    for (int s = 0; s < b->n; s++) {
        const uint32_t nbytes = b->sizes[(size_t)s];
        uint8_t head[5] = {(uint8_t)type, (uint8_t)nbytes, (uint8_t)(nbytes >> 8), (uint8_t)(nbytes >> 16), (uint8_t)(nbytes >> 24)};
        if (b->write) {   // packet header (src/enc.rs:301-305, :453-457) + payload as one write
            b->packet.assign(head, head + 5);
            b->packet.insert(b->packet.end(), b->payloads.data() + b->offsets[(size_t)s], b->payloads.data() + b->offsets[(size_t)s] + nbytes);
            b->write(b->user, s, b->packet.data(), b->packet.size());
        } else {
            be_emit(b, s, head, 5);
            be_emit(b, s, b->payloads.data() + b->offsets[(size_t)s], nbytes);
        }
    }
    return PFV_OK;
}

PFV_API void pfv_batch_encoder_destroy(pfv_batch_encoder *b)
{
    if (!b) return;
    pfv_ctx *ctx = b->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);
    for (int i = 0; i < 2; i++) {
        if (b->ev_up[i]) (void)hipEventDestroy(b->ev_up[i]);
        if (b->in_host[i]) (void)hipHostFree(b->in_host[i]);
        if (b->in_dev[i]) (void)hipFree(b->in_dev[i]);
    }
    if (b->coef) (void)hipFree(b->coef);
    if (b->mv) (void)hipFree(b->mv);
    if (b->has) (void)hipFree(b->has);
    if (b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    pfv_enc_session_destroy(b->hot);
    delete b;
}

PFV_API int pfv_batch_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, int n_streams, pfv_write_cb write,
                                     void *user, pfv_batch_encoder **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_encoder_create: bad argument");
    *out = nullptr;
    if (framerate < 0 || framerate > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "framerate must fit u16 (src/enc.rs:197)");
    pfv_enc_session *hot = nullptr;
    int rc = pfv_enc_session_create(ctx, width, height, quality, n_streams, &hot);
    if (rc) return rc;
    pfv_batch_encoder *b = new pfv_batch_encoder();
    b->ctx = ctx; b->hot = hot; b->n = n_streams; b->width = width; b->height = height;
    b->write = write; b->user = user;
    b->frame_bytes = pfv_frame_bytes(width, height);
    b->total_blocks = (size_t)pfv_total_blocks(width, height);
    b->sizes.assign((size_t)n_streams, 0);
    b->offsets.assign((size_t)n_streams, 0);
    if (!write) { b->kept.resize((size_t)n_streams); b->taken.resize((size_t)n_streams); }
    const size_t in_bytes = (size_t)n_streams * b->frame_bytes, nmb = (size_t)n_streams * b->total_blocks;
    hipError_t e = hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&b->ev_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void **)&b->in_host[i], in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&b->in_dev[i], in_bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&b->coef, nmb * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&b->mv, nmb * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&b->has, nmb);
    if (e != hipSuccess) {
        rc = hip_fail(ctx, e, "pfv_batch_encoder_create");
        pfv_batch_encoder_destroy(b);
        return rc;
    }
    rc = pfv_enc_entropy_enable(hot, 0);
    // landing zone for one step's payloads: typical content needs a fraction of the worst case; it grows on demand
    if (!rc && !b->payloads.resize(std::max<size_t>(in_bytes, 1 << 20))) rc = fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
    if (rc) { pfv_batch_encoder_destroy(b); return rc; }
    // header (src/enc.rs:190-219): magic, version, geometry, the four q-tables -- to every writer
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], nullptr);
    std::vector<uint8_t> head;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    head.insert(head.end(), magic, magic + 8);
    put_u32(head, 211);
    put_u16(head, (unsigned)width); put_u16(head, (unsigned)height); put_u16(head, (unsigned)framerate);
    put_u16(head, 4);
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 64; i++) put_u16(head, (unsigned)q[t][i]);
    for (int s = 0; s < n_streams; s++) be_emit(b, s, head.data(), head.size());
    *out = b;
    return PFV_OK;
}

// the page-locked [n_streams][frame_bytes] array to fill for the NEXT encode call (two of them alternate)
PFV_API uint8_t *pfv_batch_encoder_frames(pfv_batch_encoder *b) { return b ? b->in_host[b->step & 1] : nullptr; }

PFV_API int pfv_batch_encoder_encode(pfv_batch_encoder *b, int pframe, const uint8_t *frames)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    pfv_ctx *ctx = b->ctx;
    if (b->finished) return fail(ctx, PFV_ERR_STATE, "batch encoder already finished (src/enc.rs:80)");
    if (pframe && b->poisoned) return fail(ctx, PFV_ERR_STATE, "a previous step failed after prev_frame had advanced: encode i-frames next");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)(b->step & 1);
    const uint8_t *src = frames ? frames : b->in_host[slot];
    // in_dev[slot] was last read by the kernels of step t-2, which the collect of step t-1's call has waited for
    HIP_TRY(ctx, hipMemcpyAsync(b->in_dev[slot], src, (size_t)b->n * b->frame_bytes, hipMemcpyHostToDevice, b->copy_stream));
    // From here on the copy engine may be reading the CALLER's buffer: whatever way this call ends, it returns only once
    // that read is over ("free again when the call returns", pfv_hip.h).
    struct UploadGuard {
        hipStream_t s;
        bool armed;
        ~UploadGuard() { if (armed) (void)hipStreamSynchronize(s); }
    } guard{b->copy_stream, frames != nullptr};
    HIP_TRY(ctx, hipEventRecord(b->ev_up[slot], b->copy_stream));
    int rc = be_collect(b);            // step t-1 -> writers, while the upload of step t is on the wire
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, b->ev_up[slot], 0));
    rc = pframe ? pfv_enc_pframe_dev(b->hot, b->in_dev[slot], b->mv, b->has, b->coef) : pfv_enc_iframe_dev(b->hot, b->in_dev[slot], b->coef);
    if (rc) return rc;
    b->poisoned = true;                // until this step's packets have been written
    rc = pframe ? pfv_enc_pack_pframe_dev(b->hot, b->mv, b->has, b->coef) : pfv_enc_pack_iframe_dev(b->hot, b->coef);
    if (rc) return rc;
    if (frames) {
        guard.armed = false;
        HIP_TRY(ctx, hipStreamSynchronize(b->copy_stream));   // the caller's buffer is free again when this returns
    }
    b->pending = pframe ? 2 : 1;
    b->poisoned = false;
    b->step++;
    return PFV_OK;
}
// packets of the step in flight -> writers (encode does this for the previous step by itself)
PFV_API int pfv_batch_encoder_flush(pfv_batch_encoder *b)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    return be_collect(b);
}
PFV_API int pfv_batch_encoder_finish(pfv_batch_encoder *b)
{
    if (!b) return fail(nullptr, PFV_ERR_BAD_ARG, "null batch encoder");
    if (b->finished) return fail(b->ctx, PFV_ERR_STATE, "batch encoder already finished (src/enc.rs:183)");
    int rc = be_collect(b);
    if (rc) return rc;
    b->finished = true;
    const uint8_t eof[5] = {0, 0, 0, 0, 0};                                    // src/enc.rs:221-227
    for (int s = 0; s < b->n; s++) be_emit(b, s, eof, 5);
    return PFV_OK;
}
// write == NULL at creation: the bytes produced for one stream since the last take (valid until the next call on `b`)
PFV_API int pfv_batch_encoder_take(pfv_batch_encoder *b, int stream, const uint8_t **data, size_t *len)
{
    if (!b || !data || !len || stream < 0 || stream >= b->n || b->write) return fail(b ? b->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_batch_encoder_take: bad argument");
    b->taken[(size_t)stream].swap(b->kept[(size_t)stream]);
    b->kept[(size_t)stream].clear();
    *data = b->taken[(size_t)stream].data();
    *len = b->taken[(size_t)stream].size();
    return PFV_OK;
}

// ------------------------------------------------------------------ batch decoder: n streams, pipelined
// n .pfv streams of one geometry and one packet-type pattern (e.g. what a pfv_batch_encoder wrote) decoded together.  The
// packets of a step are bit-parsed (src/dec.rs:226-296, 328-417) on a worker pool, one task per stream, into (index, value)
// lists in page-locked memory; ONE segmented scatter kernel reads the lists straight from host memory, ONE decode launch
// serves all streams, ONE copy brings the frames back.  The parse of step t+1 runs while the device works on step t.
struct BdSet {   // host staging of one step (two sets alternate)
    PinnedBuf<uint32_t> idx;
    PinnedBuf<int16_t> val;
    PinnedBuf<uint32_t> counts;
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    std::vector<int> rc;               // per stream: 0, kSinkFull, PFV_ERR_*
    std::vector<uint8_t> qidx;         // per stream x 3
    std::vector<const uint8_t *> payload;
    std::vector<size_t> len;
    int type = 0;                      // 0 EOF, 1 i-frames, 2 p-frames, 3 drop frames; negative: error found by the scanner
    // device-entropy form of the step (PFV_OPT_ENTROPY_DECODE): what entd_prepare leaves for the k_entd_* kernels, per stream
    bool dev_form = false;
    PinnedBuf<uint8_t> bytes;          // the payloads, 16-byte aligned starts
    PinnedBuf<EdPacket> pk;            // [n]
    PinnedBuf<uint2> groups;
    std::vector<uint8_t> host_parse;   // per stream: the host parser reads this packet
    size_t bytes_total = 0;
};
struct pfv_batch_decoder {
    DecEntd entd;                      // switches, shape and counters of the device entropy stage
    DecWindow win[2];                  // its device buffers, per staging set: [n] packets, [n][total_blocks] lists / headers / coefficients
    hipStream_t win_stream = nullptr;  // the window of step t + 1 runs here, under the decode and download of step t
    pfv_ctx *ctx = nullptr;
    pfv_dec_session *hot = nullptr;
    int n = 0, width = 0, height = 0, framerate = 0, n_qtables = 0;
    size_t total_blocks = 0, frame_bytes = 0, cap = 0;
    std::vector<const uint8_t *> data;
    std::vector<size_t> len, pos;
    BdSet set[2];
    PinnedBuf<int16_t> dense;          // fallback for steps whose lists overflow
    ListStage hp;                      // device-entropy steps: a packet the device stage left to the host parser
    PinnedBuf<uint8_t> frames[2];
    uint8_t *frames_dev = nullptr;
    long step = 0, dense_steps = 0;
    bool eof = false;
    // worker pool
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    BdSet *job = nullptr;
    int next = 0, done = 0, generation = 0;
    bool quit = false;
};

static void bd_parse_one(pfv_batch_decoder *b, BdSet *s, int k)
{
    const size_t tb = b->total_blocks;
    if (s->dev_form) {   // the device reads the run streams: table, q indices, block headers and the payload's copy here
        EdPacket &pk = s->pk.data()[k];
        const EntdPrep r = entd_prepare(s->payload[(size_t)k], (uint32_t)s->len[(size_t)k], s->type, tb, b->n_qtables, b->entd.sub_bits, pk, s->bytes.data() + pk.byte_off);
        s->rc[(size_t)k] = r.rc;
        s->host_parse[(size_t)k] = r.host_parse;
        memcpy(&s->qidx[(size_t)k * 3], r.qidx, 3);
        s->counts.data()[k] = 0;
        return;
    }
    SparseSink sink{s->idx.data() + (size_t)k * b->cap, s->val.data() + (size_t)k * b->cap, b->cap};
    sink.offset = (size_t)k * tb * 256;
    uint8_t *q = &s->qidx[(size_t)k * 3];
    int rc = s->type == 2 ? parse_pframe_to(s->payload[(size_t)k], s->len[(size_t)k], (int)tb, b->n_qtables, s->mv.data() + (size_t)k * tb * 2,
                                            s->has.data() + (size_t)k * tb, sink, q)
                          : parse_iframe_to(s->payload[(size_t)k], s->len[(size_t)k], (int)tb, b->n_qtables, sink, q);
    s->counts.data()[k] = (uint32_t)sink.n;
    s->rc[(size_t)k] = rc;
}
static void bd_worker(pfv_batch_decoder *b)
{
    std::unique_lock<std::mutex> lk(b->m);
    int seen = 0;
    for (;;) {
        b->cv_work.wait(lk, [&] { return b->quit || b->generation != seen; });
        if (b->quit) return;
        seen = b->generation;
        BdSet *s = b->job;
        while (s && b->next < b->n) {
            const int k = b->next++;
            lk.unlock();
            bd_parse_one(b, s, k);
            lk.lock();
            if (++b->done == b->n) b->cv_done.notify_all();
        }
    }
}
// next frame packet of every stream (unknown packet types are skipped, src/dec.rs:216-219); starts the parse on the pool
static void bd_scan_and_start(pfv_batch_decoder *b, BdSet *s)
{
    s->type = 0;
    s->dev_form = false;        // only a step of frame packets takes the device form (set below)
    int first = -1;
    bool all_empty = true, any_empty = false;
    for (int k = 0; k < b->n; k++) {
        const uint8_t *d = b->data[(size_t)k];
        size_t p = b->pos[(size_t)k];
        int typ;
        size_t n = 0;
        for (;;) {
            if (p + 5 > b->len[(size_t)k]) { s->type = PFV_ERR_IO; return; }
            typ = d[p];
            n = (size_t)d[p + 1] | ((size_t)d[p + 2] << 8) | ((size_t)d[p + 3] << 16) | ((size_t)d[p + 4] << 24);
            if (typ == 0) break;
            if (p + 5 + n > b->len[(size_t)k]) { s->type = PFV_ERR_IO; return; }
            p += 5 + n;
            if (typ == 1 || typ == 2) break;
        }
        b->pos[(size_t)k] = p;
        if (first < 0) first = typ;
        else if (typ != first) { s->type = PFV_ERR_FORMAT; return; }     // the streams' packet types diverge at this step
        s->payload[(size_t)k] = typ ? d + p - n : nullptr;
        s->len[(size_t)k] = n;
        all_empty = all_empty && n == 0;
        any_empty = any_empty || n == 0;
    }
    if (first == 0) { s->type = 0; return; }
    if (first == 1 && all_empty) { s->type = 3; return; }                 // drop frames (src/dec.rs:188-202)
    if (any_empty) { s->type = first == 2 ? PFV_ERR_IO : PFV_ERR_FORMAT; return; }   // empty p-frame packet: truncated read (:204-214)
    s->type = first;
    s->dev_form = false;
    if (b->entd.on) {
        size_t total = 0;
        bool big = b->entd.force;
        for (int k = 0; k < b->n; k++) {
            s->pk.data()[k].byte_off = total;
            s->pk.data()[k].frame_off = (unsigned long long)k;
            total += (s->len[(size_t)k] + 16 + 15) & ~(size_t)15;
            big = big || s->len[(size_t)k] >= kDecEntdMinBytes;
        }
        s->bytes_total = total;
        s->dev_form = big && total < (1ull << 32) && s->bytes.resize(total + 64 > s->bytes.size() ? total + total / 2 + 64 : total + 64);
    }
    std::lock_guard<std::mutex> lk(b->m);
    b->job = s; b->next = 0; b->done = 0; b->generation++;
    b->cv_work.notify_all();
}
static void bd_join(pfv_batch_decoder *b, BdSet *s)
{
    if (s->type != 1 && s->type != 2) return;
    std::unique_lock<std::mutex> lk(b->m);
    while (b->job == s && b->next < b->n) {       // the caller helps (and is the whole pool when there are no workers)
        const int k = b->next++;
        lk.unlock();
        bd_parse_one(b, s, k);
        lk.lock();
        ++b->done;
    }
    b->cv_done.wait(lk, [&] { return b->done >= b->n; });
    b->job = nullptr;
}

}  // extern "C"

// The window of step s on set w (all on the window stream): payloads, packet descriptors, block headers and lists up, coefficient arrays
// cleared, k_entd_*, statuses down.
static int bd_window_enqueue(pfv_batch_decoder *b, BdSet *s, DecWindow &w)
{
    pfv_ctx *ctx = b->ctx;
    DecEntd &v = b->entd;
    const size_t S = (size_t)b->n, tb = b->total_blocks;
    int mrc = entd_windows_make(ctx, v, b->win, &b->win_stream, S, tb);
    if (mrc) return mrc;
    hipStream_t st = b->win_stream;
    size_t total_sub = 0, n_groups = 0, hdr_total = 0;
    unsigned max_hdr = 0;
    for (size_t k = 0; k < S; k++) {
        EdPacket &pk = s->pk.data()[k];
        if (s->host_parse[k] || s->rc[k]) pk.n_sub = pk.hdr_wgs = 0;
        pk.sub_first = (uint32_t)total_sub;
        pk.grp_first = (uint32_t)n_groups;
        pk.hdr_first = (uint32_t)hdr_total;
        total_sub += pk.n_sub;
        n_groups += (pk.n_sub + kEdOwn - 1) / kEdOwn;
        hdr_total += pk.hdr_wgs;
        max_hdr = std::max(max_hdr, (unsigned)pk.hdr_wgs);
    }
    if (total_sub >= 0xffffffffull) return fail(ctx, PFV_ERR_NOMEM, "batch decoder: payloads too large for one step of the device entropy stage");
    if (!s->groups.resize(n_groups + 1)) return fail(ctx, PFV_ERR_NOMEM, "pinned staging");
    {
        size_t g = 0;
        for (size_t k = 0; k < S; k++)
            for (uint32_t blk = 0; blk * (uint32_t)kEdOwn < s->pk.data()[k].n_sub; blk++) s->groups.data()[g++] = make_uint2((unsigned)k, blk);
    }
    auto room = [&](auto **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return PFV_OK;
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }               // the set is idle: its last window was consumed and decoded
        need += need / 2;
        HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(**p)));
        *cap = need;
        return PFV_OK;
    };
    int rc;
    if ((rc = room(&w.bytes_dev, &w.bytes_cap, s->bytes_total + 64))) return rc;
    if ((rc = room(&w.groups_dev, &w.groups_cap, n_groups + 1))) return rc;
    if ((rc = room(&w.sub_dev, &w.sub_cap, (total_sub + 1) * 4))) return rc;
    if ((rc = room(&w.wgsum_dev, &w.wgsum_cap, n_groups + 1))) return rc;
    if ((rc = room(&w.hdr_maps_dev, &w.hdr_maps_cap, (hdr_total + 1) * 8))) return rc;
    if ((rc = room(&w.hdr_start_dev, &w.hdr_start_cap, hdr_total + 1))) return rc;
    {   // every packet's list: its place in the window's pool from the packet's size
        size_t total = 0;
        w.list_room.assign(S, 0);
        for (size_t k = 0; k < S; k++) { w.list_room[k] = entd_pool_cap(tb, s->len[k]); total += w.list_room[k]; }
        w.lists.drop_spill();
        if ((rc = w.lists.room(ctx, total))) return rc;
        total = 0;
        for (size_t k = 0; k < S; k++) { w.lists.ptr_host.data()[k] = w.lists.ent + total; total += w.list_room[k]; }
        HIP_TRY(ctx, hipMemcpyAsync(w.lists.ptr_dev, w.lists.ptr_host.data(), S * sizeof(uint32_t *), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.bytes_dev, s->bytes.data(), s->bytes_total, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.pk_dev, s->pk.data(), S * sizeof(EdPacket), hipMemcpyHostToDevice, st));
    if (n_groups) HIP_TRY(ctx, hipMemcpyAsync(w.groups_dev, s->groups.data(), n_groups * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(w.status_dev, 0, S * sizeof(uint32_t), st));
    if (n_groups) {
        const size_t ts = w.sub_cap / 4;
        EdBufs eb{w.bytes_dev, w.pk_dev, w.groups_dev, w.sub_dev, w.sub_dev + ts, w.sub_dev + 2 * ts, w.wgsum_dev, w.coded_dev, w.lists.ptr_dev, w.lists.counts_dev, w.status_dev, 0u, 0u,
                  w.hdr_maps_dev, w.hdr_start_dev, w.mv_dev, w.has_dev};
        entd_launch(st, eb, (unsigned)S, (unsigned)n_groups, max_hdr, v.launches, v.inner);
        if ((rc = launch_check(ctx, "k_entd_*"))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.status_host.data(), w.status_dev, S * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(w.done, st));
    w.owner = (DecEvent *)s;       // an identity only: which staging set this window belongs to
    return PFV_OK;
}

extern "C" {
PFV_API void pfv_batch_decoder_destroy(pfv_batch_decoder *b)
{
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->m);
        b->quit = true;
        b->cv_work.notify_all();
    }
    for (auto &t : b->workers) t.join();
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->frames_dev) (void)hipFree(b->frames_dev);
    if (b->win_stream) { (void)hipStreamSynchronize(b->win_stream); (void)hipStreamDestroy(b->win_stream); }
    for (DecWindow &w : b->win) w.destroy();
    pfv_dec_session_destroy(b->hot);
    delete b;
}

PFV_API int pfv_batch_decoder_create(pfv_ctx *ctx, const uint8_t *const *streams, const size_t *lens, int n_streams, int n_threads,
                                     pfv_batch_decoder **out)
{
    if (!ctx || !streams || !lens || !out || n_streams <= 0 || n_threads < 0) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_decoder_create: bad argument");
    *out = nullptr;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    const uint8_t *d0 = streams[0];
    if (!d0 || lens[0] < 8) return fail(ctx, PFV_ERR_IO, "stream shorter than the magic");
    if (memcmp(d0, magic, 8) != 0) return fail(ctx, PFV_ERR_FORMAT, "bad magic (src/dec.rs:50-52)");
    if (lens[0] < 20) return fail(ctx, PFV_ERR_IO, "truncated header");
    const uint32_t ver = (uint32_t)d0[8] | ((uint32_t)d0[9] << 8) | ((uint32_t)d0[10] << 16) | ((uint32_t)d0[11] << 24);
    if (ver != 211) return fail(ctx, PFV_ERR_VERSION, "codec version is not 2.1.1 (src/dec.rs:57-59)");
    auto u16 = [&](size_t o) { return (int)d0[o] | ((int)d0[o + 1] << 8); };
    const int w = u16(12), h = u16(14), fps = u16(16), nq = u16(18);
    const size_t head = 20 + (size_t)nq * 128;
    for (int k = 0; k < n_streams; k++) {
        if (!streams[k] || lens[k] < head) return fail(ctx, PFV_ERR_IO, "truncated header");
        if (memcmp(streams[k], d0, head) != 0) return fail(ctx, PFV_ERR_FORMAT, "the streams must share one header (geometry, frame rate, q-tables)");
    }
    std::vector<int32_t> q((size_t)std::max(nq, 1) * 64, 1);
    for (int i = 0; i < nq * 64; i++) q[(size_t)i] = u16(20 + 2 * (size_t)i);
    // the coefficient lists address [stream][macroblock][256] with 32-bit flat indices (SparseSink, k_scatter_coef_seg)
    if (w > 0 && h > 0 && !(w & 1) && !(h & 1) && (uint64_t)n_streams * (uint64_t)pfv_total_blocks(w, h) * 256u > 0xffffffffull)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_batch_decoder_create: n_streams x macroblocks x 256 exceeds the 32-bit coefficient index; use several batch decoders");
    pfv_dec_session *hot = nullptr;
    int rc = pfv_dec_session_create(ctx, w, h, q.data(), nq, n_streams, &hot);
    if (rc) return rc;
    pfv_batch_decoder *b = new pfv_batch_decoder();
    b->ctx = ctx; b->hot = hot; b->n = n_streams; b->width = w; b->height = h; b->framerate = fps; b->n_qtables = nq;
    b->total_blocks = (size_t)pfv_total_blocks(w, h);
    b->frame_bytes = pfv_frame_bytes(w, h);
    b->cap = b->total_blocks * 256 / 4;                       // per stream: denser than 1 in 4 -> dense fallback
    b->data.assign(streams, streams + n_streams);
    b->len.assign(lens, lens + n_streams);
    b->pos.assign((size_t)n_streams, head);
    const size_t S = (size_t)n_streams, tb = b->total_blocks;
    bool ok = true;
    for (auto &s : b->set) {
        ok = ok && s.idx.resize(S * b->cap) && s.val.resize(S * b->cap) && s.counts.resize(S) && s.mv.resize(S * tb * 2) && s.has.resize(S * tb);
        s.rc.assign(S, 0); s.qidx.assign(S * 3, 0); s.payload.assign(S, nullptr); s.len.assign(S, 0);
    }
    ok = ok && b->frames[0].resize(S * b->frame_bytes) && b->frames[1].resize(S * b->frame_bytes);
    if (ok && ctx->opt_entropy_decode != PFV_ENTROPY_DECODE_HOST && tb > 0) {   // the steps' run streams are read on the device (big payloads; every step under _DEVICE)
        DecEntd &v = b->entd;
        v.force = ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE;
        v.sub_bits = (uint32_t)ctx->opt_entdec_lane_bits; v.launches = ctx->opt_entdec_launches; v.inner = ctx->opt_entdec_inner;
        bool host_ok = true;                // the window stream and sets: with the first step that takes the device form (bd_window_enqueue)
        for (auto &s : b->set) {
            host_ok = host_ok && s.pk.resize(S);
            s.host_parse.assign(S, 0);
        }
        v.on = host_ok;
        if (!v.on && v.force) ok = false;
    }
    hipError_t e = ok ? hipMalloc((void **)&b->frames_dev, S * b->frame_bytes) : hipErrorOutOfMemory;
    if (e == hipSuccess && (rc = dec_staging(hot)) == PFV_OK) rc = pfv_dec_set_output_dev(hot, b->frames_dev);
    if (e != hipSuccess) rc = hip_fail(ctx, e, "pfv_batch_decoder_create");
    if (rc) { pfv_batch_decoder_destroy(b); return rc; }
    for (int t = 0; t < n_threads; t++) b->workers.emplace_back(bd_worker, b);
    bd_scan_and_start(b, &b->set[0]);                         // the first step is being parsed when create returns
    *out = b;
    return PFV_OK;
}
PFV_API int pfv_batch_decoder_width(const pfv_batch_decoder *b) { return b ? b->width : 0; }
PFV_API int pfv_batch_decoder_height(const pfv_batch_decoder *b) { return b ? b->height : 0; }
PFV_API int pfv_batch_decoder_framerate(const pfv_batch_decoder *b) { return b ? b->framerate : 0; }
// steps so far whose coefficient lists overflowed (denser than 1 non-zero in 4) and went up in the dense form
PFV_API long pfv_batch_decoder_dense_steps(const pfv_batch_decoder *b) { return b ? b->dense_steps : 0; }
PFV_API void pfv_batch_decoder_entropy_counts(const pfv_batch_decoder *b, long counts_out[2])
{
    if (!b || !counts_out) return;
    counts_out[0] = b->entd.packets_dev;
    counts_out[1] = b->entd.packets_host;
}

// One step for all streams: 1 = *frames_out points at [n_streams][frame_bytes] decoded frames (page-locked, valid until the
// call after next), 2 = a step of drop frames (no frames), 0 = end of the streams, negative = error (PFV_ERR_FORMAT also when
// the streams' packet types or q-table indices diverge).
PFV_API int pfv_batch_decoder_advance(pfv_batch_decoder *b, const uint8_t **frames_out)
{
    if (!b || !frames_out) return fail(b ? b->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_batch_decoder_advance: bad argument");
    pfv_ctx *ctx = b->ctx;
    *frames_out = nullptr;
    if (b->eof) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = (int)(b->step & 1);
    BdSet *s = &b->set[slot];
    bd_join(b, s);
    if (s->type < 0) { b->eof = true; return fail(ctx, s->type, "batch decoder: truncated stream or diverging packet types"); }
    if (s->type == 0) { b->eof = true; return 0; }
    b->step++;
    if (s->type == 3) {
        bd_scan_and_start(b, &b->set[slot ^ 1]);
        return 2;
    }
    // From here on the step counter has advanced: an error exit that left the decoder usable would make the next call join a
    // slot whose contents are two steps old and decode it again as if it were new.  Every failure below ends the decoder
    // (b->eof), as the parse errors above do.
    const int rc_step = [&]() -> int {
    const size_t S = (size_t)b->n, tb = b->total_blocks;
    bool dense = false;
    for (size_t k = 0; k < S; k++) {
        if (s->rc[k] == kSinkFull) dense = true;
        else if (s->rc[k]) { b->eof = true; return fail(ctx, s->rc[k], "malformed packet payload"); }
        if (memcmp(&s->qidx[k * 3], &s->qidx[0], 3) != 0) { b->eof = true; return fail(ctx, PFV_ERR_FORMAT, "the streams use different q-table indices in this step"); }
    }
    pfv_dec_session *hot = b->hot;
    const size_t total = tb * S * 256;
    int rc = PFV_OK;
    if (s->dev_form && b->win[slot].owner != (DecEvent *)s && (rc = bd_window_enqueue(b, s, b->win[slot]))) {   // not enqueued ahead (first step, or its headers were late)
        if (b->entd.ready || b->entd.force) return rc;
        // PFV_ENTROPY_DECODE_AUTO and the window stream / sets could not be made (they are created with the first step that takes the device
        // form): the stage is switched off for this decoder, this step is parsed by the host code here and now, the ones behind it on the pool
        (void)hipGetLastError();
        b->entd.on = false;
        s->dev_form = false;
        for (size_t k = 0; k < S; k++) {
            bd_parse_one(b, s, (int)k);
            if (s->rc[k] == kSinkFull) dense = true;
            else if (s->rc[k]) { b->eof = true; return fail(ctx, s->rc[k], "malformed packet payload"); }
        }
        b->entd.packets_host += (long)S;
        rc = PFV_OK;
    }
    if (s->dev_form) {   // the step's payloads through the device's entropy stage, the host parser for what it will not take
        DecEntd &v = b->entd;
        DecWindow &w = b->win[slot];
        HIP_TRY(ctx, hipEventSynchronize(w.done));
        w.owner = nullptr;
        for (size_t k = 0; k < S; k++) {
            if (!s->host_parse[k] && !w.status_host.data()[k]) { v.packets_dev++; continue; }
            v.packets_host++;
            uint8_t q[3];
            const int prc = b->hp.parse(s->payload[k], s->len[k], s->type, tb, b->n_qtables, s->mv.data() + k * tb * 2, s->has.data() + k * tb, w.list_room[k], q);
            if (prc == PFV_ERR_NOMEM) return fail(ctx, prc, "pinned list staging");
            if (prc) { b->eof = true; return fail(ctx, prc, "malformed packet payload"); }
            if ((rc = upload_lists(ctx, w.lists, k, w.list_room[k], b->hp.ent.data(), b->hp.n, b->hp.counts.data(), ctx->stream))) return rc;
            if (s->type == 2) {     // its block headers with it (the device's read of them is not what is decoded)
                HIP_TRY(ctx, hipMemcpyAsync(w.mv_dev + k * tb * 2, s->mv.data() + k * tb * 2, tb * 2, hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(ctx, hipMemcpyAsync(w.has_dev + k * tb, s->has.data() + k * tb, tb, hipMemcpyHostToDevice, ctx->stream));
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                 // the one staging list is used again
        }
        rc = dec_step(hot, s->type == 2, w.mv_dev, w.has_dev, w.lists.coefs(), &s->qidx[0]);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(b->frames[slot].data(), b->frames_dev, S * b->frame_bytes, hipMemcpyDeviceToHost, ctx->stream));
        BdSet *nx = &b->set[slot ^ 1];
        bd_scan_and_start(b, nx);
        if (nx->dev_form) {        // its headers now (the pool and this thread, while the frames above travel), then its window on the second stream
            bd_join(b, nx);
            bool sound = true;
            for (size_t k = 0; k < S; k++) sound = sound && !nx->rc[k];
            if (sound && bd_window_enqueue(b, nx, b->win[slot ^ 1]) != PFV_OK) b->win[slot ^ 1].owner = nullptr;   // tried again when its turn comes
        }
        if ((rc = pfv_dec_check(hot))) return rc;      // synchronises; bad-motion-vector flag (src/common.rs:258-259)
        *frames_out = b->frames[slot].data();
        return 1;
    } else {
    const bool lists_on_device_bus = s->idx.pinned && s->val.pinned && s->counts.pinned;   // page-locked: the kernel can read them
    if (!dense && !lists_on_device_bus) {   // pageable staging (locked-memory limit): expand the lists on the host instead
        if (!b->dense.resize(total)) return fail(ctx, PFV_ERR_NOMEM, "dense staging");
        memset(b->dense.data(), 0, total * 2);
        for (size_t k = 0; k < S; k++)
            for (uint32_t i = 0; i < s->counts.data()[k]; i++) b->dense.data()[s->idx.data()[k * b->cap + i]] = s->val.data()[k * b->cap + i];
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_coef, b->dense.data(), total * 2, hipMemcpyHostToDevice, ctx->stream));
    } else if (dense) {   // some list overflowed (very dense content): parse every stream into the dense form on this thread
        b->dense_steps++;
        if (!b->dense.resize(total)) return fail(ctx, PFV_ERR_NOMEM, "pinned dense staging");
        memset(b->dense.data(), 0, total * 2);
        for (size_t k = 0; k < S && !rc; k++) {
            DenseSink sink{b->dense.data() + k * tb * 256};
            uint8_t q[3];
            rc = s->type == 2 ? parse_pframe_to(s->payload[k], s->len[k], (int)tb, b->n_qtables, s->mv.data() + k * tb * 2, s->has.data() + k * tb, sink, q)
                              : parse_iframe_to(s->payload[k], s->len[k], (int)tb, b->n_qtables, sink, q);
        }
        if (rc) { b->eof = true; return fail(ctx, rc, "malformed packet payload"); }
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_coef, b->dense.data(), total * 2, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIP_TRY(ctx, hipMemsetAsync(hot->st_coef, 0, total * 2, ctx->stream));
        hipLaunchKernelGGL(k_scatter_coef_seg, dim3(64, (unsigned)S), dim3(kThreads), 0, ctx->stream, s->idx.data(), s->val.data(),
                           s->counts.data(), (uint32_t)b->cap, (uint32_t)total, hot->st_coef);
        if ((rc = launch_check(ctx, "k_scatter_coef_seg"))) return rc;
    }
    }
    if (s->type == 2) {
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_mv, s->mv.data(), S * tb * 2, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_has, s->has.data(), S * tb, hipMemcpyHostToDevice, ctx->stream));
        rc = pfv_dec_pframe_dev(hot, hot->st_mv, hot->st_has, hot->st_coef, &s->qidx[0]);
    } else {
        rc = pfv_dec_iframe_dev(hot, hot->st_coef, &s->qidx[0]);
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(b->frames[slot].data(), b->frames_dev, S * b->frame_bytes, hipMemcpyDeviceToHost, ctx->stream));
    bd_scan_and_start(b, &b->set[slot ^ 1]);       // parse of step t+1 under the device work of step t
    if ((rc = pfv_dec_check(hot))) return rc;      // synchronises; bad-motion-vector flag (src/common.rs:258-259)
    *frames_out = b->frames[slot].data();
    return 1;
    }();
    if (rc_step < 0) b->eof = true;
    return rc_step;
}

}  // extern "C"
