// pfv_decoder_object.hip -- payload serialisers / parsers on their own, and pfv_decoder (dec::Decoder<R>) with its look-ahead machinery.
// Part of the one translation unit of the C ABI: included by pfv_capi.hip, in this order, never compiled on its own.
extern "C" {

// payload serialisers alone (for tests: product vs oracle on identical coefficient input)
PFV_API size_t pfv_serialize_iframe_payload(const int16_t *coef, int total_blocks, uint8_t *out, size_t cap)
{
    std::vector<uint8_t> p;
    if (!coef || total_blocks <= 0 || !serialize_iframe(p, coef, total_blocks)) return 0;
    if (out && p.size() <= cap) memcpy(out, p.data(), p.size());
    return p.size();
}
PFV_API size_t pfv_serialize_pframe_payload(const int8_t *mv, const uint8_t *has_coef, const int16_t *coef, int total_blocks,
                                            uint8_t *out, size_t cap)
{
    std::vector<uint8_t> p;
    if (!mv || !has_coef || !coef || total_blocks <= 0 || !serialize_pframe(p, mv, has_coef, coef, total_blocks)) return 0;
    if (out && p.size() <= cap) memcpy(out, p.data(), p.size());
    return p.size();
}

// payload parsers alone (decode_iframe / decode_pframe up to the plane decode, src/dec.rs:226-296, 328-417); host only.
// coef_out: [total_blocks][256], zero-filled first.  Returns PFV_OK, PFV_ERR_FORMAT or PFV_ERR_IO.
PFV_API int pfv_parse_iframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int16_t *coef_out,
                                     uint8_t qidx_out[3])
{
    if (!payload || !coef_out || !qidx_out || total_blocks <= 0) return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_iframe_payload: bad argument");
    int rc = parse_iframe(payload, len, total_blocks, n_qtables, coef_out, qidx_out);
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}
PFV_API int pfv_parse_pframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int8_t *mv_out,
                                     uint8_t *has_coef_out, int16_t *coef_out, uint8_t qidx_out[3])
{
    if (!payload || !mv_out || !has_coef_out || !coef_out || !qidx_out || total_blocks <= 0)
        return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_pframe_payload: bad argument");
    int rc = parse_pframe(payload, len, total_blocks, n_qtables, mv_out, has_coef_out, coef_out, qidx_out);
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}
// The sparse form the stream decoder uploads: up to `cap` (flat index, value) pairs; *n_out = pairs written.  Returns 1 when
// the list would overflow (the caller then parses the dense form).
PFV_API int pfv_parse_payload_sparse(int is_pframe, const uint8_t *payload, size_t len, int total_blocks, int n_qtables,
                                     int8_t *mv_out, uint8_t *has_coef_out, uint32_t *idx_out, int16_t *val_out, size_t cap,
                                     size_t *n_out, uint8_t qidx_out[3])
{
    if (!payload || !idx_out || !val_out || !n_out || !qidx_out || total_blocks <= 0 || (is_pframe && (!mv_out || !has_coef_out)))
        return fail(nullptr, PFV_ERR_BAD_ARG, "pfv_parse_payload_sparse: bad argument");
    SparseSink sink{idx_out, val_out, cap};
    int rc = is_pframe ? parse_pframe_to(payload, len, total_blocks, n_qtables, mv_out, has_coef_out, sink, qidx_out)
                       : parse_iframe_to(payload, len, total_blocks, n_qtables, sink, qidx_out);
    *n_out = sink.n;
    if (rc == kSinkFull) return 1;
    return rc ? fail(nullptr, rc, "malformed packet payload") : PFV_OK;
}

// Decoder::new (src/dec.rs:38-134).  `data` must stay valid for the decoder's lifetime (R: Read + Seek).
PFV_API int pfv_decoder_create(pfv_ctx *ctx, const uint8_t *data, size_t len, pfv_decoder **out)
{
    if (!ctx || !data || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_decoder_create: bad argument");
    *out = nullptr;
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    if (len < 8) return fail(ctx, PFV_ERR_IO, "stream shorter than the magic (DecodeError::IOError)");
    if (memcmp(data, magic, 8) != 0) return fail(ctx, PFV_ERR_FORMAT, "bad magic (DecodeError::FormatError, src/dec.rs:50-52)");
    if (len < 12) return fail(ctx, PFV_ERR_IO, "truncated header");
    uint32_t ver = (uint32_t)data[8] | ((uint32_t)data[9] << 8) | ((uint32_t)data[10] << 16) | ((uint32_t)data[11] << 24);
    if (ver != 211) return fail(ctx, PFV_ERR_VERSION, "codec version is not 2.1.1 (DecodeError::VersionError, src/dec.rs:57-59)");
    if (len < 20) return fail(ctx, PFV_ERR_IO, "truncated header");
    auto u16 = [&](size_t o) { return (int)data[o] | ((int)data[o + 1] << 8); };
    int w = u16(12), h = u16(14), fps = u16(16), nq = u16(18);
    if (len < 20 + (size_t)nq * 128) return fail(ctx, PFV_ERR_IO, "truncated q-tables");
    std::vector<int32_t> q((size_t)std::max(nq, 1) * 64, 1);
    for (int i = 0; i < nq * 64; i++) q[i] = u16(20 + 2 * (size_t)i);
    pfv_dec_session *hot = nullptr;
    int rc = pfv_dec_session_create(ctx, w, h, q.data(), nq, 1, &hot);
    if (rc) return rc;
    pfv_decoder *d = new pfv_decoder();
    d->ctx = ctx; d->hot = hot; d->data = data; d->len = len;
    d->pos = d->reset_pos = d->scan_pos = 20 + (size_t)nq * 128;
    d->width = w; d->height = h; d->framerate = fps; d->n_qtables = nq;
    d->total_blocks = pfv_total_blocks(w, h);
    if (!d->retframe.resize(pfv_frame_bytes(w, h))) {
        pfv_decoder_destroy(d);
        return fail(ctx, PFV_ERR_NOMEM, "pfv_decoder_create: pinned staging");
    }
    memset(d->retframe.data(), 0, (size_t)w * h);                              // VideoFrame::new (frame.rs:12-26): Y 0, U/V 128
    memset(d->retframe.data() + (size_t)w * h, 128, d->retframe.size() - (size_t)w * h);
    if (ctx->opt_entropy_decode != PFV_ENTROPY_DECODE_HOST && d->total_blocks > 0) {   // the run streams of big packets are read on the device
        DecEntd &v = d->entd;
        v.force = ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE;
        v.sub_bits = (uint32_t)ctx->opt_entdec_lane_bits; v.launches = ctx->opt_entdec_launches; v.inner = ctx->opt_entdec_inner;
        v.on = true;                     // the window stream and sets: with the first packet that takes the device form (dec_window_enqueue)
    }
    const unsigned hw = std::thread::hardware_concurrency();
    if ((rc = pfv_decoder_set_lookahead(d, hw > 1 ? (int)std::min(4u, hw - 1) : 0))) {
        pfv_decoder_destroy(d);
        return rc;
    }
    *out = d;
    return PFV_OK;
}
}  // extern "C"

// ---- look-ahead machinery of pfv_decoder
static void dec_parse(pfv_decoder *d, DecEvent *e)   // any thread; touches only the event and the immutable stream
{
    const size_t tb = (size_t)d->total_blocks, cap = tb * 256 / 4;   // denser than 1 in 4: not worth a list
    e->dense = false;
    e->n_sparse = 0;
    e->dev_form = e->host_parse = false;
    if (d->entd.on && (d->entd.force || e->plen >= kDecEntdMinBytes)) {   // the device reads the run streams: only the headers here
        const uint32_t max_sub = (uint32_t)(((uint64_t)e->plen * 8 + d->entd.sub_bits - 1) / d->entd.sub_bits);
        if (!e->bytes.resize((size_t)e->plen + 32) || !e->pk.resize(1) || !e->groups.resize((size_t)max_sub / kEdOwn + 1) ||
            (e->type == 2 && (!e->mv.resize(tb * 2) || !e->has.resize(tb)))) {
            e->rc = PFV_ERR_NOMEM;
            return;
        }
        EdPacket &k = *e->pk.data();
        k.byte_off = 0; k.frame_off = 0;
        const EntdPrep r = entd_prepare(e->payload, e->plen, e->type, tb, d->n_qtables, d->entd.sub_bits, k, e->bytes.data());
        e->rc = r.rc;
        memcpy(e->qidx, r.qidx, 3);
        e->dev_form = true;
        e->host_parse = r.host_parse;
        if (r.rc || r.host_parse) k.n_sub = k.hdr_wgs = 0;
        const uint32_t ng = (k.n_sub + kEdOwn - 1) / kEdOwn;
        for (uint32_t g = 0; g < ng; g++) e->groups.data()[g] = make_uint2(0u, g);
        if (!e->rc && e->host_parse) {   // the host parser decides about this one, here, on this thread
            if (!e->coef.resize(tb * 256)) { e->rc = PFV_ERR_NOMEM; return; }
            e->rc = e->type == 1 ? parse_iframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->coef.data(), e->qidx)
                                 : parse_pframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(), e->coef.data(), e->qidx);
            e->dev_form = false;
            e->dense = true;
        }
        return;
    }
    if (!e->idx.resize(cap) || !e->val.resize(cap) || (e->type == 2 && (!e->mv.resize(tb * 2) || !e->has.resize(tb)))) {
        e->rc = PFV_ERR_NOMEM;
        return;
    }
    SparseSink sink{e->idx.data(), e->val.data(), cap};
    e->rc = e->type == 1 ? parse_iframe_to(e->payload, e->plen, d->total_blocks, d->n_qtables, sink, e->qidx)
                         : parse_pframe_to(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(), sink,
                                           e->qidx);
    e->n_sparse = sink.n;
    if (e->rc != kSinkFull) return;
    e->dense = true;
    if (!e->coef.resize(tb * 256)) {
        e->rc = PFV_ERR_NOMEM;
        return;
    }
    e->rc = e->type == 1 ? parse_iframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->coef.data(), e->qidx)
                         : parse_pframe(e->payload, e->plen, d->total_blocks, d->n_qtables, e->mv.data(), e->has.data(),
                                        e->coef.data(), e->qidx);
}
static void dec_worker(pfv_decoder *d)
{
    (void)hipSetDevice(d->ctx->device);   // the pinned landing zones are allocated from this thread
    std::unique_lock<std::mutex> lk(d->m);
    for (;;) {
        DecEvent *job = nullptr;
        for (size_t k = 0; k < d->count && !job; k++) {
            DecEvent *e = d->ring[(d->head + k) % d->ring.size()].get();
            if (e->state == DecEvent::QUEUED) job = e;
        }
        if (d->quit) return;
        if (!job) { d->cv_work.wait(lk); continue; }
        job->state = DecEvent::RUNNING;
        lk.unlock();
        dec_parse(d, job);
        lk.lock();
        job->state = DecEvent::DONE;
        d->cv_done.notify_all();
    }
}
// Walks packet headers from scan_pos exactly as the reference's loop would (dec.rs:174-222) and queues what it finds
// until the ring is full or an END / ERROR event is pending.  Caller holds the lock.
static void dec_scan(pfv_decoder *d)
{
    bool queued = false;
    while (d->count < d->ring.size() && !d->scan_stop) {
        DecEvent *e = d->ring[(d->head + d->count) % d->ring.size()].get();
        size_t pos = d->scan_pos;
        auto emit = [&](DecEvent::Kind kind, DecEvent::State st, size_t pos_after) {
            e->kind = kind; e->state = st; e->pos_after = pos_after;
            d->count++;
        };
        if (pos + 5 > d->len) {
            e->rc = PFV_ERR_IO; e->msg = "unexpected end of stream in a packet header";
            emit(DecEvent::ERROR, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        const uint8_t type = d->data[pos];
        const uint32_t plen = (uint32_t)d->data[pos + 1] | ((uint32_t)d->data[pos + 2] << 8) | ((uint32_t)d->data[pos + 3] << 16) |
                              ((uint32_t)d->data[pos + 4] << 24);
        pos += 5;
        if (type == 0) {   // EOF marker (:183-187)
            emit(DecEvent::END, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        if (pos + plen > d->len) {
            e->rc = PFV_ERR_IO; e->msg = "packet payload runs past the end of the stream";
            emit(DecEvent::ERROR, DecEvent::DONE, pos);
            d->scan_stop = true;
            break;
        }
        const uint8_t *payload = d->data + pos;
        pos += plen;
        d->scan_pos = pos;
        if (type != 1 && type != 2) continue;   // unknown packet: skipped (:216-219)
        if (type == 1 && plen == 0) {           // drop frame: nothing decoded, no callback (:190)
            emit(DecEvent::DROP, DecEvent::DONE, pos);
            continue;
        }
        e->type = type; e->payload = payload; e->plen = plen; e->rc = 0;
        emit(DecEvent::FRAME, DecEvent::QUEUED, pos);
        queued = true;
    }
    if (queued) d->cv_work.notify_all();
}
// Forget everything scanned ahead and continue from `pos`.  Caller holds the lock.
static void dec_rewind(pfv_decoder *d, std::unique_lock<std::mutex> &lk, size_t pos)
{
    for (;;) {   // a parse in flight keeps pointers into its event: let it finish
        bool running = false;
        for (auto &e : d->ring) running |= e->state == DecEvent::RUNNING;
        if (!running) break;
        d->cv_done.wait(lk);
    }
    if (d->win_stream) (void)hipStreamSynchronize(d->win_stream);          // a window enqueued ahead reads its event's buffers
    for (DecWindow &w : d->win) w.owner = nullptr;
    for (auto &e : d->ring) e->state = DecEvent::FREE;
    d->head = d->count = 0;
    d->scan_pos = d->pos = pos;
    d->scan_stop = false;
}
static void dec_stop_workers(pfv_decoder *d)
{
    {
        std::lock_guard<std::mutex> lk(d->m);
        d->quit = true;
    }
    d->cv_work.notify_all();
    for (auto &t : d->workers) t.join();
    d->workers.clear();
    d->quit = false;
}

extern "C" {

// Packets parsed ahead of the one being decoded, on `n_threads` worker threads (0: parse inline, no threads).  The
// default is min(4, hardware threads - 1).  Frames, order and error codes are those of the sequential loop.
PFV_API int pfv_decoder_set_lookahead(pfv_decoder *d, int n_threads)
{
    if (!d || n_threads < 0 || n_threads > 64) return fail(d ? d->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_decoder_set_lookahead: bad argument");
    dec_stop_workers(d);
    std::unique_lock<std::mutex> lk(d->m);
    dec_rewind(d, lk, d->pos);
    d->ring.clear();
    for (int i = 0; i < std::max(n_threads + 1, kDecWindows); i++) d->ring.emplace_back(new DecEvent());     // kDecWindows at least: the packets behind the current one are scanned (and, without threads, prepared by the caller's thread)
    lk.unlock();
    for (int i = 0; i < n_threads; i++) d->workers.emplace_back(dec_worker, d);
    return PFV_OK;
}

PFV_API void pfv_decoder_destroy(pfv_decoder *d)
{
    if (!d) return;
    dec_stop_workers(d);
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->win_stream) { (void)hipStreamSynchronize(d->win_stream); (void)hipStreamDestroy(d->win_stream); }
    if (d->frame_dev) (void)hipFree(d->frame_dev);
    for (DecWindow &w : d->win) w.destroy();
    pfv_dec_session_destroy(d->hot);
    delete d;
}
// on != 0: the decoded frame stays in device memory and the callback's y / u / v are DEVICE pointers to the packed frame (valid until the
// next advance call) -- for consumers on the GPU; the frame's download, more than half of a 4K call, is not paid
PFV_API int pfv_decoder_set_output_device(pfv_decoder *d, int on)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    pfv_ctx *ctx = d->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !d->frame_dev) HIP_TRY(ctx, hipMalloc((void **)&d->frame_dev, pfv_frame_bytes(d->width, d->height)));
    if (!on && d->frame_dev) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(d->frame_dev); d->frame_dev = nullptr; }
    return PFV_OK;
}
PFV_API void pfv_decoder_entropy_counts(const pfv_decoder *d, long counts_out[2])
{
    if (!d || !counts_out) return;
    counts_out[0] = d->entd.packets_dev;
    counts_out[1] = d->entd.packets_host;
}
PFV_API int pfv_decoder_width(const pfv_decoder *d) { return d ? d->width : 0; }          // dec.rs:136-138
PFV_API int pfv_decoder_height(const pfv_decoder *d) { return d ? d->height : 0; }        // dec.rs:140-142
PFV_API int pfv_decoder_framerate(const pfv_decoder *d) { return d ? d->framerate : 0; }  // dec.rs:144-146
// Decoder::reset (src/dec.rs:148-152)
PFV_API int pfv_decoder_reset(pfv_decoder *d)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    std::unique_lock<std::mutex> lk(d->m);
    d->eof = false;
    dec_rewind(d, lk, d->reset_pos);
    return PFV_OK;
}

}  // extern "C"

// The window of packet e on set w: uploads, cleared coefficient array, k_entd_*, status download -- all on the window stream.
static int dec_window_enqueue(pfv_decoder *d, DecEvent *e, DecWindow &w)
{
    pfv_ctx *ctx = d->ctx;
    DecEntd &v = d->entd;
    const size_t tb = (size_t)d->total_blocks;
    int mrc = entd_windows_make(ctx, v, d->win, &d->win_stream, 1, tb);
    if (mrc) return mrc;
    hipStream_t st = d->win_stream;
    const EdPacket &k = *e->pk.data();
    const uint32_t ng = (k.n_sub + kEdOwn - 1) / kEdOwn;
    auto room = [&](auto **p, size_t *cap, size_t need) -> int {
        if (need <= *cap) return PFV_OK;
        if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }           // the set is idle: its last window was consumed and decoded
        need += need / 2;
        HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(**p)));
        *cap = need;
        return PFV_OK;
    };
    int rc;
    if ((rc = room(&w.bytes_dev, &w.bytes_cap, (size_t)e->plen + 64))) return rc;
    if ((rc = room(&w.groups_dev, &w.groups_cap, (size_t)ng + 1))) return rc;
    if ((rc = room(&w.sub_dev, &w.sub_cap, ((size_t)k.n_sub + 1) * 4))) return rc;
    if ((rc = room(&w.wgsum_dev, &w.wgsum_cap, (size_t)ng + 1))) return rc;
    if ((rc = room(&w.hdr_maps_dev, &w.hdr_maps_cap, ((size_t)k.hdr_wgs + 1) * 8))) return rc;
    if ((rc = room(&w.hdr_start_dev, &w.hdr_start_cap, (size_t)k.hdr_wgs + 1))) return rc;
    w.list_room.assign(1, entd_pool_cap(tb, e->plen));
    w.lists.drop_spill();
    if ((rc = w.lists.room(ctx, w.list_room[0]))) return rc;
    w.lists.ptr_host.data()[0] = w.lists.ent;
    HIP_TRY(ctx, hipMemcpyAsync(w.lists.ptr_dev, w.lists.ptr_host.data(), sizeof(uint32_t *), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.bytes_dev, e->bytes.data(), (size_t)e->plen + 16, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(w.pk_dev, e->pk.data(), sizeof(EdPacket), hipMemcpyHostToDevice, st));
    if (ng) HIP_TRY(ctx, hipMemcpyAsync(w.groups_dev, e->groups.data(), ng * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(w.status_dev, 0, sizeof(uint32_t), st));
    if (ng) {
        const size_t ts = w.sub_cap / 4;
        EdBufs b{w.bytes_dev, w.pk_dev, w.groups_dev, w.sub_dev, w.sub_dev + ts, w.sub_dev + 2 * ts, w.wgsum_dev, w.coded_dev, w.lists.ptr_dev, w.lists.counts_dev, w.status_dev, 0u, 0u,
                 w.hdr_maps_dev, w.hdr_start_dev, w.mv_dev, w.has_dev};
        entd_launch(st, b, 1u, ng, k.hdr_wgs, v.launches, v.inner);
        if ((rc = launch_check(ctx, "k_entd_*"))) return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(w.status_host.data(), w.status_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(w.done, st));
    w.owner = e;
    return PFV_OK;
}
// One packet through the device's entropy stage, then the decode launch; before the frame is fetched, the window of the
// packet behind it -- if its headers are ready -- is put on the window stream, where it runs under this frame's decode and download.
static int dec_consume_entd(pfv_decoder *d, DecEvent *e)
{
    pfv_ctx *ctx = d->ctx;
    pfv_dec_session *hot = d->hot;
    DecEntd &v = d->entd;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t tb = (size_t)d->total_blocks;
    int rc;
    DecWindow *w = nullptr;
    for (DecWindow &x : d->win)
        if (x.owner == e) w = &x;
    if (!w) {                     // not enqueued ahead: now (every window is free or holds a packet behind this one; the last call ended with the stream idle)
        for (DecWindow &x : d->win)
            if (!w && !x.owner) w = &x;
        if (!w) { w = &d->win[0]; w->owner = nullptr; }
        if ((rc = dec_window_enqueue(d, e, *w))) { w->owner = nullptr; return rc; }
    }
    HIP_TRY(ctx, hipEventSynchronize(w->done));
    w->owner = nullptr;           // consumed (event objects are reused by the ring: a stale match would take this window for a later packet's)
    if (*w->status_host.data()) {   // the device stage is not certain about this payload: the host parser reads it and decides
        v.packets_host++;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // the staging list's last upload
        const int prc = d->hp.parse(e->payload, e->plen, e->type, tb, d->n_qtables, e->mv.data(), e->has.data(), w->list_room[0], e->qidx);
        if (prc) return fail(ctx, prc, prc == PFV_ERR_NOMEM ? "pinned list staging" : "malformed packet payload");
        if ((rc = upload_lists(ctx, w->lists, 0, w->list_room[0], d->hp.ent.data(), d->hp.n, d->hp.counts.data(), ctx->stream))) return rc;
        if (e->type == 2) {         // its block headers with it (the device's read of them is not what is decoded)
            HIP_TRY(ctx, hipMemcpyAsync(w->mv_dev, e->mv.data(), tb * 2, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(w->has_dev, e->has.data(), tb, hipMemcpyHostToDevice, ctx->stream));
        }
    } else {
        v.packets_dev++;
    }
    rc = dec_step(hot, e->type == 2, w->mv_dev, w->has_dev, w->lists.coefs(), e->qidx);
    if (rc) return rc;
    // The packets behind this one: their windows (payload upload, k_hdr_*, k_entd_*, status) go onto the window stream now, where they run
    // under this frame's decode and download -- up to kDecWindows - 1 of them: one packet's window is a chain of a dozen small kernels
    // (~0.15 ms of latency for a 4K p-frame), so with a single window ahead the chain of packet t + 1 only started when packet t's decode was
    // launched and every frame waited for most of it.  The window just consumed is still being read by the decode launched above: it is not
    // among the free ones until this call has returned.
    for (size_t k = 1; k < (size_t)kDecWindows; k++) {
        std::unique_lock<std::mutex> lk(d->m);
        DecEvent *nx = d->count > k ? d->ring[(d->head + k) % d->ring.size()].get() : nullptr;
        if (!nx) break;
        if (nx->state == DecEvent::QUEUED && d->workers.empty()) {
            // no parser threads (pfv_decoder_set_lookahead(d, 0)): this thread reads the packet's first 19 bytes and stages its payload now,
            // while the decode just launched runs -- with the block headers read on the device that is all a big packet needs from the host
            nx->state = DecEvent::RUNNING;
            lk.unlock();
            dec_parse(d, nx);
            lk.lock();
            nx->state = DecEvent::DONE;
        }
        const bool ready = nx->state == DecEvent::DONE && nx->kind == DecEvent::FRAME && nx->dev_form && !nx->rc;
        const bool stop = nx->state != DecEvent::DONE || nx->kind == DecEvent::END || nx->kind == DecEvent::ERROR;
        lk.unlock();
        if (stop) break;
        if (!ready) continue;
        DecWindow *free_w = nullptr;
        bool has = false;
        for (DecWindow &x : d->win) {
            has = has || x.owner == nx;
            if (!free_w && !x.owner && &x != w) free_w = &x;
        }
        if (has) continue;
        if (!free_w) break;
        if (dec_window_enqueue(d, nx, *free_w) != PFV_OK) { free_w->owner = nullptr; break; }   // it will be tried again when its turn comes
    }
    return pfv_dec_check(hot);
}

extern "C" {
// Decoder::advance_frame (src/dec.rs:169-224).  Returns 1 = Ok(true), 0 = Ok(false) (EOF), negative = error.
// onvideo(user, y, u, v, width, height) is called for every decoded frame (not for drop frames).
PFV_API int pfv_decoder_advance_frame(pfv_decoder *d, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    if (d->eof) return 0;
    std::unique_lock<std::mutex> lk(d->m);
    dec_scan(d);
    DecEvent *e = d->ring[d->head].get();
    while (e->state != DecEvent::DONE) {
        if (e->state == DecEvent::QUEUED) {   // nobody picked it up yet: parse it here
            e->state = DecEvent::RUNNING;
            lk.unlock();
            dec_parse(d, e);
            lk.lock();
            e->state = DecEvent::DONE;
        } else {
            d->cv_done.wait(lk);
        }
    }
    // consume the event; the slot stays reserved (FREE but not rescanned) until the device has read its buffers
    d->pos = e->pos_after;
    const DecEvent::Kind kind = e->kind;
    if (kind == DecEvent::END || kind == DecEvent::ERROR) {
        const int rc = e->rc;
        const char *msg = e->msg;
        dec_rewind(d, lk, d->pos);   // nothing was scanned past it; the next call rescans from pos like the reference
        if (kind == DecEvent::END) {
            d->eof = true;
            return 0;
        }
        return fail(d->ctx, rc, msg);
    }
    int rc = PFV_OK;
    if (kind == DecEvent::FRAME) {
        lk.unlock();   // workers keep parsing the packets behind this one while the device decodes it
        rc = e->rc;
        if (rc) rc = fail(d->ctx, rc, rc == PFV_ERR_NOMEM ? "pinned staging for a parsed packet" : "malformed packet payload");
        if (!rc && e->host_parse && !e->dev_form) d->entd.packets_host++;   // a packet of device size the host parser had to read (degenerate table, 64 MiB and more)
        if (!rc && e->dev_form) {
            rc = dec_consume_entd(d, e);
            if (rc && !d->entd.ready && !d->entd.force) {
                // PFV_ENTROPY_DECODE_AUTO and the window stream / sets could not be made (they are created with the first packet that takes the
                // device form): the device stage is switched off for this decoder and the host parser reads this packet -- and the ones the
                // parser threads have already prepared in device form, each when its turn comes.  An error only under PFV_ENTROPY_DECODE_DEVICE.
                (void)hipGetLastError();
                d->entd.on = false;
                dec_parse(d, e);
                rc = e->rc;
                if (rc) rc = fail(d->ctx, rc, rc == PFV_ERR_NOMEM ? "pinned staging for a parsed packet" : "malformed packet payload");
                else d->entd.packets_host++;
            }
        }
        if (rc || e->dev_form)
            ;
        else if (e->dense)
            rc = e->type == 1 ? pfv_dec_iframe(d->hot, e->coef.data(), e->qidx)
                              : pfv_dec_pframe(d->hot, e->mv.data(), e->has.data(), e->coef.data(), e->qidx);
        else
            rc = e->type == 1 ? pfv_dec_iframe_sparse(d->hot, e->idx.data(), e->val.data(), e->n_sparse, e->qidx)
                              : pfv_dec_pframe_sparse(d->hot, e->mv.data(), e->has.data(), e->idx.data(), e->val.data(),
                                                      e->n_sparse, e->qidx);
        if (!rc && d->frame_dev) {   // pfv_decoder_set_output_device: the retframe stays in device memory
            rc = pfv_dec_get_frame_dev(d->hot, d->frame_dev);
            if (!rc) rc = pfv_ctx_sync(d->ctx);
        } else if (!rc) {
            rc = pfv_dec_get_frame(d->hot, d->retframe.data());   // crop blits (:195-197, 209-211)
        }
        lk.lock();
    }
    e->state = DecEvent::FREE;
    d->head = (d->head + 1) % d->ring.size();
    d->count--;
    dec_scan(d);       // refill the freed slot right away
    lk.unlock();
    if (rc) return rc;
    if (kind == DecEvent::FRAME && onvideo) {
        size_t ny = (size_t)d->width * d->height, nc = (size_t)(d->width / 2) * (d->height / 2);
        const uint8_t *f = d->frame_dev ? d->frame_dev : d->retframe.data();
        onvideo(user, f, f + ny, f + ny + nc, d->width, d->height);
    }
    return 1;
}

// Decoder::advance_delta (src/dec.rs:154-167)
PFV_API int pfv_decoder_advance_delta(pfv_decoder *d, double delta, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    d->delta_accum += delta;
    double delta_per_frame = 1.0 / (double)d->framerate;
    while (d->delta_accum >= delta_per_frame) {
        int rc = pfv_decoder_advance_frame(d, onvideo, user);
        if (rc <= 0) return rc;
        d->delta_accum -= delta_per_frame;
    }
    return 1;
}

}  // extern "C"
