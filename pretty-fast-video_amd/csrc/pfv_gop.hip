// pfv_gop.hip -- GOP-batched stream objects (included by pfv_capi.hip; uses its sessions, PinnedBuf and the host parsers).
//
// enc::Encoder / dec::Decoder (src/enc.rs:12-188, src/dec.rs:15-224) for ONE stream, with the independent GOPs of the stream as the
// slots of every launch.  encode_iframe never reads prev_frame and overwrites all three planes of it (src/enc.rs:84-97);
// decode_plane_into overwrites the whole framebuffer (src/common.rs:477-496): the runs I P P ... of a stream share nothing, so frame
// t of EVERY run of a batch goes through one launch per stage instead of one launch per frame.  A single 4K stream then fills the
// device like 20 streams do (bench.py --workload config5: 0.97 G -> 1.39 G macroblocks/s at kernel scope).  The bytes written and the
// frames delivered are those of the frame-by-frame objects (pfv_encoder / pfv_decoder); only WHEN they appear differs: a packet
// leaves when its batch is complete.
//
//   batch        up to max_gops runs ("groups") of up to max_gop_frames frames; a group starts at an i-frame.  A run longer than
//                max_gop_frames continues in slot 0 of the next batch (its reference frame is carried over, one device copy), and so
//                does a stream that starts with p-frames (prev_frame = new_padded, src/enc.rs:46).
//   frame step   t = 0 .. longest group - 1: the slots whose group has a frame t, as maximal runs of neighbouring slots of one frame
//                type (normally ONE launch: all groups are equally long but the last).  The ping-pong index of the session
//                flips once per step; a slot that sits a step out never reads its stale side (its next frame is an i-frame, or the
//                state is copied explicitly: failed packets on the decoder side).
#pragma once

// wall-clock accounting of where a GOP-batched object spends its host time (pfv_gop_*_stats): cheap (two clock reads per section)
struct GopClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

// PFV_GOP_TRACE=1: a host-side log of the encoder's steps (seconds since the object was created) on stderr -- where a rocprofv3 trace would
// distort the host's own timing
struct GopTrace {
    bool on = getenv("PFV_GOP_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()(const char *what, long a = 0, long b = 0) const
    {
        if (on) fprintf(stderr, "[pfv gop %9.3f ms] %s %ld %ld\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3, what, a, b);
    }
};

struct GopPacket {
    uint8_t type;     // 1 i-frame, 2 p-frame, 3 drop frame (src/enc.rs:175-180)
    int slot, t;
};

struct GopEncBatch {
    std::vector<int> len;              // frames per group; slot = index
    std::vector<uint8_t> first_type;   // 1: the group starts with an i-frame; 2: it continues a run (or the stream starts with p-frames)
    std::vector<GopPacket> order;      // packets of the batch in stream order
    bool in_flight = false;            // kernels enqueued, payloads not yet collected
    int steps = 0;                     // frame steps enqueued (longest group)
    uint8_t *frames_dev = nullptr;     // [max_gop_frames][max_gops][frame_bytes]
    // where each frame of the batch lies: its place in frames_dev, or -- device frames taken BY REFERENCE -- the caller's own buffer
    PinnedBuf<const uint8_t *> slots_host;      // [max_gop_frames][max_gops]
    const uint8_t **slots_dev = nullptr;
    bool by_ref = false;               // some frame of the batch is read where the caller left it
    uint8_t *arena = nullptr;          // retained payloads of the batch
    size_t arena_cap = 0;              // its size: grows (this batch's arena alone) after a batch outgrew it
    uint8_t *cont_save = nullptr;      // the reference frame slot 0 continued from, saved at submit (one padded frame; allocated with the first such batch):
    bool cont_saved = false;           //   what a batch that outgrew its arena is encoded again from (gop_enc_redo)
    std::vector<uint8_t> redo;         // the packets' payloads of such a batch, made again frame by frame; the pending segments point into it
    EntEntry *entries_dev = nullptr;   // [max_gop_frames][max_gops]
    unsigned long long *cursor_dev = nullptr;
    hipEvent_t ev_uploaded = nullptr, ev_done = nullptr, ev_dev_frames = nullptr;
    bool dev_frames = false;           // frames were copied on the caller's stream: the batch's kernels wait for ev_dev_frames
    // the payloads come over step by step, under the kernels of the steps behind them: after step t the arena's fill level is copied to
    // cursor_steps[t] (page-locked) and ev_step[t] recorded; whoever next looks at the batch (any encode call, the collection) fetches the
    // bytes the finished steps added (gop_enc_fetch)
    std::vector<hipEvent_t> ev_step;
    PinnedBuf<unsigned long long> cursor_steps;
    int steps_fetched = 0;
    size_t fetched_bytes = 0;
    PinnedBuf<uint8_t> payload_host;   // the batch's payloads on the host (page-locked): the pending segments point into it
    std::vector<uint8_t> heads;        // 5 bytes per packet of the batch
    void clear() { len.clear(); first_type.clear(); order.clear(); in_flight = false; dev_frames = false; by_ref = false; steps = 0; steps_fetched = 0; fetched_bytes = 0; }
    int frames() const { int n = 0; for (int l : len) n += l; return n; }
};

struct pfv_gop_encoder {
    // ctx: the encoder's OWN launch context (a stream of its own for the batches' kernels); user: the context the caller created the encoder
    // on -- frames that lie in device memory are copied on ITS stream (the *_dev ordering), so the copies of the batch being filled run under
    // the kernels of the batch in flight instead of queueing behind them
    pfv_ctx *ctx = nullptr;      // ctx->owner = the caller's context, or nullptr once that has been destroyed
    pfv_enc_session *hot = nullptr;
    int width = 0, height = 0, max_gops = 0, max_len = 0;
    size_t frame_bytes = 0, total_blocks = 0, arena_cap = 0;       // arena_cap: what a batch's arena starts with
    bool explicit_budget = false;          // the caller gave pfv_gop_encoder_create a payload budget: outgrowing it is an error, not a reason to grow
    int quality = 0;
    long batches_redone = 0;
    std::vector<uint32_t> redo_sizes;      // payload bytes of the packets of the batch made again, in stream order (drop frames left out)
    GopEncBatch batch[2];
    int cur = 0;                           // batch being filled
    hipStream_t copy_stream = nullptr;     // plane uploads
    hipStream_t down_stream = nullptr;     // payload downloads (its own stream: an upload's wait must not queue behind them)
    int16_t *coef = nullptr;               // encode outputs of one step, max_gops wide
    int8_t *mv = nullptr;
    uint8_t *has = nullptr;
    bool cont_valid = false;               // a group is open across the batch boundary: where its prev_frame lives
    int cont_buf = 0, cont_slot = 0;
    PinnedBuf<EntEntry> entries_host;
    unsigned long long *cursor_host = nullptr;   // page-locked
    // the writer side: bytes produced and not yet handed over = `out` (contiguous) followed by `segs` (packet headers and payloads where
    // they lie: a batch's payloads stay in its page-locked landing zone until the batch slot is collected again)
    std::vector<uint8_t> out, drained;
    std::vector<pfv_iovec> segs, segs_drained;
    unsigned segs_in = 0;                  // bit b: pending segments point into batch[b]'s landing zone / header bytes
    bool finished = false, failed = false;
    bool frames_by_ref = false;            // pfv_gop_encoder_set_frames_by_reference
    long frames_in = 0, batches = 0, frames_by_reference = 0;
    // seconds: [0] waiting for plane uploads, [1] enqueueing batches, [2] waiting for a batch's kernels, [3] payloads device -> host,
    // [4] packet assembly
    double stats[5] = {0, 0, 0, 0, 0};
    GopTrace trace;
};

// ---- helpers of both objects
template <class F>
static void gop_runs(const std::vector<int> &key, F &&fn)   // maximal runs of equal non-negative keys over neighbouring slots
{
    const int n = (int)key.size();
    for (int a = 0; a < n;) {
        if (key[(size_t)a] < 0) { a++; continue; }
        int b = a + 1;
        while (b < n && key[(size_t)b] == key[(size_t)a]) b++;
        fn(a, b - a, key[(size_t)a]);
        a = b;
    }
}

static void gop_put_header(std::vector<uint8_t> &o, int width, int height, int framerate, int quality)
{
    int32_t q[4][64];
    pfv_qtables_from_quality(quality, q[0], q[1], q[2], q[3], nullptr);
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};      // common.rs:1
    o.insert(o.end(), magic, magic + 8);
    put_u32(o, 211);                                                           // common.rs:2
    put_u16(o, (unsigned)width); put_u16(o, (unsigned)height); put_u16(o, (unsigned)framerate);
    put_u16(o, 4);
    for (int t = 0; t < 4; t++)                                                // intra_l, intra_c, inter_l, inter_c (enc.rs:199-215)
        for (int i = 0; i < 64; i++) put_u16(o, (unsigned)q[t][i]);
}

// the options of the caller's context as they stand now (the launches read them from the encoder's own)
static void gop_enc_take_options(pfv_gop_encoder *e)
{
    pfv_ctx *k = e->ctx;
    const pfv_ctx *u = e->ctx->owner;
    if (!u) return;          // the caller's context is gone: the options stay as they were last taken
    k->opt_enc_transform = u->opt_enc_transform;
    k->opt_tile_compaction = u->opt_tile_compaction;
    k->opt_lane_mapping = u->opt_lane_mapping;
}

// every frame step of a batch, enqueued without a host round trip
static int gop_enc_submit(pfv_gop_encoder *e, GopEncBatch &B)
{
    pfv_ctx *ctx = e->ctx;
    pfv_enc_session *s = e->hot;
    const int G = (int)B.len.size();
    if (G == 0 || B.in_flight) return PFV_OK;
    GopClock clk;
    e->trace("submit begin: batch, groups", (long)(&B - e->batch), G);
    gop_enc_take_options(e);
    HIP_TRY(ctx, hipEventRecord(B.ev_uploaded, e->copy_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, B.ev_uploaded, 0));
    if (B.dev_frames) {
        if (!ctx->owner) return fail(ctx, PFV_ERR_STATE, "the context the encoder was created on has been destroyed: its device-frame copies have no stream");
        HIP_TRY(ctx, hipEventRecord(B.ev_dev_frames, ctx->owner->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, B.ev_dev_frames, 0));
    }
    const size_t pad = (size_t)s->geom.pad_frame_bytes;
    B.cont_saved = false;
    if (B.first_type[0] == 2 && e->cont_valid) {   // slot 0 continues the run the previous batch left open: carry its reference frame over
        const uint8_t *src = s->prev[e->cont_buf] + (size_t)e->cont_slot * pad;
        uint8_t *dst = s->prev[s->cur];
        if (src != dst) HIP_TRY(ctx, hipMemcpyAsync(dst, src, pad, hipMemcpyDeviceToDevice, ctx->stream));
        if (!e->explicit_budget) {                 // and keep a copy: what the batch is encoded again from should it outgrow its arena
            if (!B.cont_save) HIP_TRY(ctx, hipMalloc((void **)&B.cont_save, pad));
            HIP_TRY(ctx, hipMemcpyAsync(B.cont_save, src, pad, hipMemcpyDeviceToDevice, ctx->stream));
            B.cont_saved = true;
        }
    }
    HIP_TRY(ctx, hipMemsetAsync(B.cursor_dev, 0, sizeof(unsigned long long), ctx->stream));
    if (B.by_ref)    // the batch's table of frame pointers (page-locked -> device, ahead of the kernels on their stream)
        HIP_TRY(ctx, hipMemcpyAsync(B.slots_dev, B.slots_host.data(), (size_t)e->max_len * (size_t)e->max_gops * sizeof(const uint8_t *), hipMemcpyHostToDevice, ctx->stream));
    int steps = 0;
    for (int l : B.len) steps = std::max(steps, l);
    const int cur0 = s->cur;
    std::vector<int> key((size_t)G);
    EntFrame f{};
    f.cap_bytes = s->ent_cap;
    for (int t = 0; t < steps; t++) {
        for (int k = 0; k < G; k++) key[(size_t)k] = B.len[(size_t)k] > t ? (t == 0 ? B.first_type[(size_t)k] : 2) : -1;
        const uint8_t *frames_t = B.frames_dev + (size_t)t * (size_t)e->max_gops * e->frame_bytes;
        int rc = PFV_OK;
        gop_runs(key, [&](int first, int count, int type) {
            if (!rc) rc = enc_launch(s, type == 2, first, count, frames_t, e->mv, e->has, e->coef, B.by_ref ? B.slots_dev + (size_t)t * (size_t)e->max_gops : nullptr);
            if (!rc) rc = ent_pack_win(s, type == 2, first, count, e->mv, e->has, e->coef);
            if (rc) return;
            EntEntry *ent = B.entries_dev + (size_t)t * (size_t)e->max_gops + (size_t)first;
            EntBufs b = s->ent;
            b.sizes += first;
            b.payload += (size_t)first * (size_t)s->ent_cap;
            hipLaunchKernelGGL(k_ent_retain, dim3(1), dim3(64), 0, ctx->stream, b.sizes, count, B.cursor_dev, (unsigned long long)B.arena_cap, ent, B.cursor_steps.data() + t);
            f.n_streams = count;
            hipLaunchKernelGGL(k_ent_gather_entries, dim3(32, (unsigned)count), dim3(kEntThreads), 0, ctx->stream, f, b, ent, B.arena);
            rc = launch_check(ctx, "k_ent_retain / k_ent_gather_entries");
        });
        if (rc) return rc;
        s->cur ^= 1;
        HIP_TRY(ctx, hipEventRecord(B.ev_step[(size_t)t], ctx->stream));
    }
    // the last group may go on in the next batch: its reference frame is in the buffer its last step wrote
    e->cont_valid = true;
    e->cont_slot = G - 1;
    e->cont_buf = (cur0 + B.len[(size_t)G - 1]) & 1;
    HIP_TRY(ctx, hipEventRecord(B.ev_done, ctx->stream));
    B.steps = steps;
    B.steps_fetched = 0;
    B.fetched_bytes = 0;
    B.in_flight = true;
    e->batches++;
    e->stats[1] += clk.lap();
    e->trace("submit end: batch, steps", (long)(&B - e->batch), steps);
    return PFV_OK;
}

// pending segments -> the contiguous byte vector (callers that did not take them before their buffers are needed again, and the
// contiguous drain / bytes calls)
static void gop_enc_materialize(pfv_gop_encoder *e)
{
    size_t n = 0;
    for (const pfv_iovec &v : e->segs) n += v.len;
    e->out.reserve(e->out.size() + n);
    for (const pfv_iovec &v : e->segs) e->out.insert(e->out.end(), v.data, v.data + v.len);
    e->segs.clear();
    e->segs_in = 0;
}

// The payload bytes the finished steps of an in-flight batch added to its arena: device -> the batch's landing zone, on the download stream.
// wait: every step (the batch is being collected); otherwise only the steps whose event has fired (called from the encode calls, so that
// the bytes travel under the kernels of the steps and the batch behind them instead of after the last kernel).
static int gop_enc_fetch(pfv_gop_encoder *e, GopEncBatch &B, bool wait)
{
    pfv_ctx *ctx = e->ctx;
    if (!B.in_flight) return PFV_OK;
    if (B.steps_fetched < B.steps && (e->segs_in & (1u << (unsigned)(&B - e->batch)))) gop_enc_materialize(e);   // segments of the batch's previous use that nobody took yet
    while (B.steps_fetched < B.steps) {
        hipEvent_t ev = B.ev_step[(size_t)B.steps_fetched];
        if (wait) HIP_TRY(ctx, hipEventSynchronize(ev));
        else if (hipEventQuery(ev) != hipSuccess) { (void)hipGetLastError(); break; }
        const size_t upto = std::min((size_t)B.cursor_steps.data()[B.steps_fetched], B.arena_cap);
        if (upto > B.payload_host.size()) {
            // the landing zone is too small (page-locking is slow: it grows in big steps): what has arrived moves to the new one
            HIP_TRY(ctx, hipStreamSynchronize(e->down_stream));
            PinnedBuf<uint8_t> bigger;
            if (!bigger.resize(std::min(B.arena_cap, upto + upto / 2 + ((size_t)4 << 20)))) return fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
            memcpy(bigger.data(), B.payload_host.data(), B.fetched_bytes);
            B.payload_host.swap(bigger);
        }
        if (upto > B.fetched_bytes) {
            HIP_TRY(ctx, hipMemcpyAsync(B.payload_host.data() + B.fetched_bytes, B.arena + B.fetched_bytes, upto - B.fetched_bytes, hipMemcpyDeviceToHost, e->down_stream));
        }
        if (e->trace.on) (void)hipLaunchHostFunc(e->down_stream, [](void *p) { (*(const GopTrace *)p)("   ... a download arrived (stream 1)"); }, &e->trace);
        e->trace(wait ? "download issued (waited): step, bytes" : "download issued (polled): step, bytes", B.steps_fetched, (long)(upto - std::min(upto, B.fetched_bytes)));
        B.fetched_bytes = std::max(B.fetched_bytes, upto);
        B.steps_fetched++;
    }
    return PFV_OK;
}

// A batch whose payloads outgrew its arena (default budget): every packet of it is made again, one frame at a time, on a one-stream session of
// its own -- the frames still lie where the batch read them (its frame array, or the caller's buffers under the by-reference contract), a
// group starts with an i-frame or, slot 0, from the reference frame saved at submit.  Same arithmetic, same bytes; serial and slow, which is
// fine for what is at most a once-per-content event: the batch's arena then grows so that the batches behind it fit.
static int gop_enc_redo(pfv_gop_encoder *e, GopEncBatch &B)
{
    pfv_ctx *ctx = e->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the batch behind this one may be running: it shares coef / mv / has with the pass below
    HIP_TRY(ctx, hipStreamSynchronize(e->down_stream));
    pfv_enc_session *r = nullptr;
    int rc = pfv_enc_session_create(ctx, e->width, e->height, e->quality, 1, &r);
    if (!rc) rc = pfv_enc_entropy_enable(r, 0);
    B.redo.clear();
    e->redo_sizes.clear();
    const size_t pad = (size_t)e->hot->geom.pad_frame_bytes;
    int last_slot = -1;
    for (size_t i = 0; !rc && i < B.order.size(); i++) {
        const GopPacket &p = B.order[i];
        if (p.type == 3) continue;
        if (p.slot != last_slot && p.type == 2) {             // a group that starts with a p-frame: slot 0 continuing the run before the batch
            if (B.cont_saved) rc = hipMemcpyAsync((void *)pfv_enc_prev_frame_dev(r, 0), B.cont_save, pad, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? PFV_OK : hip_fail(ctx, hipGetLastError(), "gop_enc_redo");
            // (no saved frame: the stream itself starts with p-frames, and a new session's prev_frame is the state they start from, src/frame.rs:38-43)
        }
        last_slot = p.slot;
        const uint8_t *f = B.slots_host.data()[(size_t)p.t * (size_t)e->max_gops + (size_t)p.slot];
        if (!rc) rc = p.type == 1 ? pfv_enc_iframe_dev(r, f, e->coef) : pfv_enc_pframe_dev(r, f, e->mv, e->has, e->coef);
        if (!rc) rc = p.type == 1 ? pfv_enc_pack_iframe_dev(r, e->coef) : pfv_enc_pack_pframe_dev(r, e->mv, e->has, e->coef);
        uint32_t size = 0;
        if (!rc) rc = pfv_enc_payload_sizes(r, &size);          // synchronises; PFV_ERR_FORMAT for a coefficient of more than 15 size bits
        if (rc) break;
        const size_t at = B.redo.size();
        B.redo.resize(at + size);
        if (size) rc = pfv_enc_payload_fetch(r, 0, B.redo.data() + at, size);
        e->redo_sizes.push_back(size);
    }
    pfv_enc_session_destroy(r);
    if (rc) return rc;
    e->batches_redone++;
    // this batch's arena for the batches to come: one and a half times what the batch needed (every payload on a 16-byte boundary)
    const size_t need = B.redo.size() + 16 * e->redo_sizes.size();
    const size_t want = ((need + need / 2) + 15) & ~(size_t)15;
    if (want > B.arena_cap) {
        uint8_t *bigger = nullptr;
        if (hipMalloc((void **)&bigger, want) == hipSuccess) {
            (void)hipFree(B.arena);
            B.arena = bigger;
            B.arena_cap = want;
        } else {
            (void)hipGetLastError();                            // it stays as it is; the next batch that outgrows it is made again like this one
        }
    }
    return PFV_OK;
}

// wait for a batch, bring its payloads over and write its packets in stream order
static int gop_enc_collect(pfv_gop_encoder *e, GopEncBatch &B)
{
    pfv_ctx *ctx = e->ctx;
    const unsigned slot_bit = 1u << (unsigned)(&B - e->batch);
    if (e->segs_in & slot_bit) gop_enc_materialize(e);   // this slot's landing zone is about to be overwritten: segments nobody took yet
    //                                                     are copied out first
    if (!B.in_flight) {   // nothing was encoded: only drop frames can be pending
        gop_enc_materialize(e);
        for (const GopPacket &p : B.order)
            if (p.type == 3) put_packet(e->out, 1, nullptr);
        B.clear();
        return PFV_OK;
    }
    const size_t n_ent = (size_t)B.steps * (size_t)e->max_gops;
    // the batch is complete on the device; its results come over on the copy stream (idle: every upload was waited for), NOT behind
    // the kernels of the next batch, which may already be queued on the context's stream
    GopClock clk;
    {   // the steps' payloads as they complete (most have come over already, under the kernels), then the batch's entry table
        const int frc = gop_enc_fetch(e, B, true);
        if (frc) { e->failed = true; return frc; }
    }
    HIP_TRY(ctx, hipEventSynchronize(B.ev_done));
    e->stats[2] += clk.lap();
    e->trace("batch done on the device: batch", (long)(&B - e->batch));
    HIP_TRY(ctx, hipMemcpyAsync(e->entries_host.data(), B.entries_dev, n_ent * sizeof(EntEntry), hipMemcpyDeviceToHost, e->down_stream));
    HIP_TRY(ctx, hipStreamSynchronize(e->down_stream));
    int rc = PFV_OK;
    for (const GopPacket &p : B.order) {
        if (p.type == 3) continue;
        const uint32_t sz = e->entries_host.data()[(size_t)p.t * (size_t)e->max_gops + (size_t)p.slot].size;
        if (sz == kEntErrOversize) rc = PFV_ERR_FORMAT;
        else if (sz == kEntErrCapacity && rc == PFV_OK) rc = PFV_ERR_NOMEM;
    }
    if (rc == PFV_ERR_NOMEM && !e->explicit_budget) {
        // the batch outgrew its arena and nobody asked for a bound: Encoder::encode_pframe cannot fail for size (src/enc.rs:125-173), so the
        // batch's packets are made again, frame by frame, and this arena grows for the batches to come
        rc = gop_enc_redo(e, B);
        if (rc) { e->failed = true; return rc; }
        e->stats[3] += clk.lap();
        B.heads.resize(B.order.size() * 5);
        e->segs_in |= slot_bit;
        size_t hi = 0, k = 0, off = 0;
        for (const GopPacket &p : B.order) {      // packets in stream order: 5 header bytes, then the payload where gop_enc_redo left it
            uint8_t *h = &B.heads[hi];
            hi += 5;
            if (p.type == 3) { h[0] = 1; h[1] = h[2] = h[3] = h[4] = 0; e->segs.push_back(pfv_iovec{h, 5}); continue; }
            const uint32_t size = e->redo_sizes[k++];
            h[0] = p.type; h[1] = (uint8_t)size; h[2] = (uint8_t)(size >> 8); h[3] = (uint8_t)(size >> 16); h[4] = (uint8_t)(size >> 24);
            e->segs.push_back(pfv_iovec{h, 5});
            if (size) e->segs.push_back(pfv_iovec{B.redo.data() + off, (size_t)size});
            off += size;
        }
        e->stats[4] += clk.lap();
        B.clear();
        return PFV_OK;
    }
    if (rc) {
        e->failed = true;
        return fail(ctx, rc, rc == PFV_ERR_FORMAT ? "coefficient needs more than 15 size bits (src/rle.rs:44)"
                                                  : "the batch's packet payloads exceed the payload budget given to pfv_gop_encoder_create");
    }
    e->stats[3] += clk.lap();
    e->trace("payloads on the host: batch, bytes", (long)(&B - e->batch), (long)B.fetched_bytes);
    // packets in stream order as segments: 5 header bytes (src/enc.rs:301-305, :453-457), then the payload where it lies
    B.heads.resize(B.order.size() * 5);
    e->segs_in |= slot_bit;
    size_t hi = 0;
    for (const GopPacket &p : B.order) {
        uint8_t *h = &B.heads[hi];
        hi += 5;
        if (p.type == 3) {                                                // drop frame: an empty i-frame packet (src/enc.rs:175-180)
            h[0] = 1; h[1] = h[2] = h[3] = h[4] = 0;
            e->segs.push_back(pfv_iovec{h, 5});
            continue;
        }
        const EntEntry &en = e->entries_host.data()[(size_t)p.t * (size_t)e->max_gops + (size_t)p.slot];
        h[0] = p.type; h[1] = (uint8_t)en.size; h[2] = (uint8_t)(en.size >> 8); h[3] = (uint8_t)(en.size >> 16); h[4] = (uint8_t)(en.size >> 24);
        e->segs.push_back(pfv_iovec{h, 5});
        if (en.size) e->segs.push_back(pfv_iovec{B.payload_host.data() + en.offset, (size_t)en.size});
    }
    e->stats[4] += clk.lap();
    B.clear();
    return PFV_OK;
}

// the batch being filled is complete: enqueue it, turn to the other one (collecting what it still holds)
static int gop_enc_rotate(pfv_gop_encoder *e)
{
    int rc = gop_enc_submit(e, e->batch[e->cur]);
    if (rc) { e->failed = true; return rc; }
    e->cur ^= 1;
    return gop_enc_collect(e, e->batch[e->cur]);
}

static int gop_enc_frame_inner(pfv_gop_encoder *e, int type, const uint8_t *y, const uint8_t *u, const uint8_t *v, bool on_device);
static int gop_enc_frame(pfv_gop_encoder *e, int type, const uint8_t *y, const uint8_t *u, const uint8_t *v, bool on_device = false)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    pfv_ctx *ctx = e->ctx;
    if (!y || !u || !v) return fail(ctx, PFV_ERR_BAD_ARG, "null plane");
    if (e->finished) return fail(ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:80)");
    if (e->failed) return fail(ctx, PFV_ERR_STATE, "an earlier batch failed: the stream is incomplete");
    const int rc = gop_enc_frame_inner(e, type, y, u, v, on_device);
    if (rc) e->failed = true;          // a frame is missing from the stream from here on (the reference's Encoder would have panicked)
    return rc;
}
static int gop_enc_frame_inner(pfv_gop_encoder *e, int type, const uint8_t *y, const uint8_t *u, const uint8_t *v, bool on_device)
{
    pfv_ctx *ctx = e->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GopEncBatch *B = &e->batch[e->cur];
    int rc = PFV_OK;
    if (type == 1) {
        if ((int)B->len.size() == e->max_gops) { if ((rc = gop_enc_rotate(e))) return rc; B = &e->batch[e->cur]; }
        B->len.push_back(0); B->first_type.push_back(1);
    } else if (B->len.empty() || B->len.back() == e->max_len) {
        // a p-frame with no open group in this batch: the run continues from the previous batch (or the stream starts with p-frames)
        if (!B->len.empty()) { if ((rc = gop_enc_rotate(e))) return rc; B = &e->batch[e->cur]; }
        B->len.push_back(0); B->first_type.push_back(2);
    }
    if ((rc = gop_enc_fetch(e, e->batch[e->cur ^ 1], false))) return rc;      // the batch in flight: the payloads of its finished steps start travelling
    const int slot = (int)B->len.size() - 1, t = B->len.back()++;
    // the three planes go straight to their place in the step's frame array (VideoFrame, src/frame.rs:3-9: no packing on the host)
    uint8_t *dst = B->frames_dev + ((size_t)t * (size_t)e->max_gops + (size_t)slot) * e->frame_bytes;
    B->slots_host.data()[(size_t)t * (size_t)e->max_gops + (size_t)slot] = dst;
    const size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;   // *_dev: a frame that is in device memory already
    if (on_device) {
        // ordered on the stream of the CALLER's context like every *_dev call of the library (behind whatever produced the frame there, ahead of
        // whatever overwrites it there), and without a host wait: the batch's kernels -- on the encoder's own stream -- wait for an event
        // recorded behind the batch's last copy.  (Round 4 copied on the upload stream and waited for it per frame: 300 waits were a third
        // of the 22 ms a 300-frame 4K clip took; until the encoder had a stream of its own the copies queued behind the previous batch's kernels.)
        if (!ctx->owner) return fail(ctx, PFV_ERR_STATE, "the context the encoder was created on has been destroyed (pfv_gop_encoder_encode_*_dev copies on its stream)");
        if (e->frames_by_ref && ((uintptr_t)y & 15) == 0) {
            // by reference: no copy at all -- the step's kernels read the frame where it lies (FrameGeom::src_slots).  The caller keeps it
            // valid and unchanged until the batch has been collected (pfv_hip_ext.h: pfv_gop_encoder_set_frames_by_reference)
            B->slots_host.data()[(size_t)t * (size_t)e->max_gops + (size_t)slot] = y;
            B->by_ref = true;
            e->frames_by_reference++;
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(dst, y, ny + 2 * nc, kind, ctx->owner->stream));
        }
        B->dev_frames = true;
        B->order.push_back(GopPacket{(uint8_t)type, slot, t});
        e->frames_in++;
        return PFV_OK;
    }
    if (u == y + ny && v == u + nc) {   // a packed frame: one copy
        HIP_TRY(ctx, hipMemcpyAsync(dst, y, ny + 2 * nc, kind, e->copy_stream));
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(dst, y, ny, kind, e->copy_stream));
        HIP_TRY(ctx, hipMemcpyAsync(dst + ny, u, nc, kind, e->copy_stream));
        HIP_TRY(ctx, hipMemcpyAsync(dst + ny + nc, v, nc, kind, e->copy_stream));
    }
    B->order.push_back(GopPacket{(uint8_t)type, slot, t});
    e->frames_in++;
    // the caller's planes are free again when the call returns (they are being read by the copy engine until then; the kernels of the
    // previous batch run underneath)
    GopClock clk;
    HIP_TRY(ctx, hipStreamSynchronize(e->copy_stream));
    e->stats[0] += clk.lap();
    return PFV_OK;
}

extern "C" {

PFV_API void pfv_gop_encoder_destroy(pfv_gop_encoder *e)
{
    if (!e) return;
    pfv_ctx *ctx = e->ctx;
    (void)hipSetDevice(ctx->device);
    if (ctx->owner) (void)hipStreamSynchronize(ctx->owner->stream);      // frame copies into the batch buffers
    (void)hipStreamSynchronize(ctx->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    for (GopEncBatch &B : e->batch) {
        if (B.entries_dev) (void)hipFree(B.entries_dev);
        if (B.frames_dev) (void)hipFree(B.frames_dev);
        if (B.slots_dev) (void)hipFree(B.slots_dev);
        if (B.arena) (void)hipFree(B.arena);
        if (B.cont_save) (void)hipFree(B.cont_save);
        if (B.cursor_dev) (void)hipFree(B.cursor_dev);
        if (B.ev_uploaded) (void)hipEventDestroy(B.ev_uploaded);
        if (B.ev_done) (void)hipEventDestroy(B.ev_done);
        if (B.ev_dev_frames) (void)hipEventDestroy(B.ev_dev_frames);
        for (hipEvent_t ev : B.ev_step) (void)hipEventDestroy(ev);
    }
    if (e->down_stream) { (void)hipStreamSynchronize(e->down_stream); (void)hipStreamDestroy(e->down_stream); }
    if (e->coef) (void)hipFree(e->coef);
    if (e->mv) (void)hipFree(e->mv);
    if (e->has) (void)hipFree(e->has);
    if (e->cursor_host) (void)hipHostFree(e->cursor_host);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    pfv_enc_session_destroy(e->hot);
    pfv_ctx_destroy(ctx);
    delete e;
}

// Encoder::new (src/enc.rs:37-73) + the batch shape.  max_gops: groups per batch = slots per launch; max_gop_frames: frames a group may
// have inside one batch (a longer run continues in the next batch); payload_budget: bytes of device memory for the packet payloads of
// ONE batch.  0 (default): twice the batch's raw frame bytes (at least 16 MiB) -- real content stays below 1.6 x raw (binary noise at
// quality 0: 1.52 x); a batch that outgrows its arena all the same is encoded again frame by frame (gop_enc_redo) and the arena grows, so like
// Encoder::encode_pframe (src/enc.rs:125-173) the object cannot fail for size.  An explicit budget is kept as given: a batch whose payloads
// exceed it fails with PFV_ERR_NOMEM and the stream stays incomplete (the caller asked for the bound).
PFV_API int pfv_gop_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, int max_gops, int max_gop_frames,
                                   size_t payload_budget, pfv_gop_encoder **out)
{
    if (!ctx || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_gop_encoder_create: bad argument");
    *out = nullptr;
    if (framerate < 0 || framerate > 65535) return fail(ctx, PFV_ERR_BAD_ARG, "framerate must fit u16 (src/enc.rs:197)");
    if (max_gops <= 0 || max_gop_frames <= 0 || max_gops > 4096 || max_gop_frames > 4096)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_gop_encoder_create: max_gops and max_gop_frames must be in 1..4096");
    pfv_ctx *user = ctx;
    int rc = ctx_create_child(user, &ctx);              // from here on `ctx` is the encoder's own launch context (the caller's device and stream priority)
    if (rc) return fail(user, rc, pfv_last_error(nullptr));
    pfv_enc_session *hot = nullptr;
    rc = pfv_enc_session_create(ctx, width, height, quality, max_gops, &hot);
    if (rc) { pfv_ctx_destroy(ctx); return rc; }
    pfv_gop_encoder *e = new pfv_gop_encoder();
    e->ctx = ctx; e->hot = hot; e->width = width; e->height = height; e->max_gops = max_gops; e->max_len = max_gop_frames;
    e->frame_bytes = pfv_frame_bytes(width, height);
    e->total_blocks = (size_t)pfv_total_blocks(width, height);
    const size_t cap_frames = (size_t)max_gops * (size_t)max_gop_frames, nmb = (size_t)max_gops * e->total_blocks;
    // default: twice the batch's raw frame bytes -- real content stays below 1.6 x (binary noise at quality 0: 1.52 x); a batch that outgrows
    // its arena all the same is encoded again frame by frame and the arena grows (gop_enc_redo): the object cannot fail for size.  An explicit
    // budget is kept as given.  PFV_TEST_GOP_ARENA_BYTES (tests only): a default small enough for ordinary content to outgrow.
    e->explicit_budget = payload_budget != 0;
    e->quality = quality;
    e->arena_cap = payload_budget ? payload_budget : std::max<size_t>(2 * cap_frames * e->frame_bytes, (size_t)16 << 20);
    if (!payload_budget && getenv("PFV_TEST_GOP_ARENA_BYTES")) e->arena_cap = std::max<size_t>(64, strtoull(getenv("PFV_TEST_GOP_ARENA_BYTES"), nullptr, 10));
    e->arena_cap = (e->arena_cap + 15) & ~(size_t)15;
    hipError_t he = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->down_stream, hipStreamNonBlocking);
    for (GopEncBatch &B : e->batch) {
        for (int t = 0; t < max_gop_frames && he == hipSuccess; t++) {
            hipEvent_t ev = nullptr;
            he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (he == hipSuccess) B.ev_step.push_back(ev);
        }
        if (he == hipSuccess && (!B.cursor_steps.resize((size_t)max_gop_frames) || !B.cursor_steps.pinned)) he = hipErrorOutOfMemory;   // k_ent_retain stores to it
        if (he == hipSuccess) he = hipMalloc((void **)&B.frames_dev, cap_frames * e->frame_bytes);
        if (he == hipSuccess) he = hipMalloc((void **)&B.slots_dev, cap_frames * sizeof(const uint8_t *));
        if (he == hipSuccess && (!B.slots_host.resize(cap_frames) || !B.slots_host.pinned)) he = hipErrorOutOfMemory;
        if (he == hipSuccess) { he = hipMalloc((void **)&B.arena, e->arena_cap); B.arena_cap = e->arena_cap; }
        if (he == hipSuccess) he = hipMalloc((void **)&B.entries_dev, cap_frames * sizeof(EntEntry));
        if (he == hipSuccess) he = hipMalloc((void **)&B.cursor_dev, sizeof(unsigned long long));
        if (he == hipSuccess) he = hipEventCreateWithFlags(&B.ev_uploaded, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&B.ev_done, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&B.ev_dev_frames, hipEventDisableTiming);
    }
    if (he == hipSuccess) he = hipMalloc((void **)&e->coef, nmb * 512);
    if (he == hipSuccess) he = hipMalloc((void **)&e->mv, nmb * 2);
    if (he == hipSuccess) he = hipMalloc((void **)&e->has, nmb);
    if (he == hipSuccess) he = hipHostMalloc((void **)&e->cursor_host, sizeof(unsigned long long), hipHostMallocDefault);
    if (he != hipSuccess) {
        rc = hip_fail(ctx, he, "pfv_gop_encoder_create");
        pfv_gop_encoder_destroy(e);
        return rc;
    }
    rc = pfv_enc_entropy_enable(hot, 0);
    if (!rc && !e->entries_host.resize(cap_frames)) rc = fail(ctx, PFV_ERR_NOMEM, "pinned staging");
    // landing zones for the payloads of a batch: a sixth of its raw bytes (+ 16 KiB) to begin with (quality-5 p-frames of noisy content
    // reach a tenth); they grow on demand
    for (GopEncBatch &B : e->batch)
        if (!rc && !B.payload_host.resize(std::min(e->arena_cap, cap_frames * e->frame_bytes / 6 + ((size_t)16 << 10)))) rc = fail(ctx, PFV_ERR_NOMEM, "pinned payload staging");
    if (rc) { pfv_gop_encoder_destroy(e); return rc; }
    gop_put_header(e->out, width, height, framerate, quality);               // write_header (src/enc.rs:190-219)
    *out = e;
    return PFV_OK;
}

// Encoder::encode_iframe / encode_pframe / encode_dropframe (src/enc.rs:75-123, 125-173, 175-180).  The planes may be reused as soon
// as the call returns; the packet appears (pfv_gop_encoder_drain) when its batch is complete -- pfv_gop_encoder_flush forces that.
PFV_API int pfv_gop_encoder_encode_iframe(pfv_gop_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v) { return gop_enc_frame(e, 1, y, u, v); }
PFV_API int pfv_gop_encoder_encode_pframe(pfv_gop_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v) { return gop_enc_frame(e, 2, y, u, v); }
// the same for a packed frame (Y | U | V, pfv_frame_bytes) that lies in DEVICE memory -- frames a renderer or another kernel left in HBM:
// nothing crosses PCIe on the way in.  Ordered on the context's stream like every *_dev call: the frame is read behind the work enqueued
// there before the call and may be overwritten by work enqueued there after it (a producer on another stream: pfv_ctx_wait_event).
static int gop_enc_frame_dev(pfv_gop_encoder *e, int type, const uint8_t *frame_dev)
{
    if (!e || !frame_dev) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_gop_encoder_encode_*_dev: bad argument");
    const size_t ny = (size_t)e->width * e->height, nc = (size_t)(e->width / 2) * (e->height / 2);
    return gop_enc_frame(e, type, frame_dev, frame_dev + ny, frame_dev + ny + nc, true);
}
// Frames handed to the *_dev calls are read WHERE THEY LIE instead of being copied into the batch (16-byte aligned frames; others are copied as
// before).  The caller promises that such a frame stays valid and unchanged until its batch has been collected: until the packet of that frame
// has been handed out (pfv_gop_encoder_drain*), or pfv_gop_encoder_flush / _finish has returned.  Same bytes.
PFV_API int pfv_gop_encoder_set_frames_by_reference(pfv_gop_encoder *e, int on)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    e->frames_by_ref = on != 0;
    return PFV_OK;
}
PFV_API int pfv_gop_encoder_encode_iframe_dev(pfv_gop_encoder *e, const uint8_t *frame_dev) { return gop_enc_frame_dev(e, 1, frame_dev); }
PFV_API int pfv_gop_encoder_encode_pframe_dev(pfv_gop_encoder *e, const uint8_t *frame_dev) { return gop_enc_frame_dev(e, 2, frame_dev); }
PFV_API int pfv_gop_encoder_encode_dropframe(pfv_gop_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:176)");
    if (e->failed) return fail(e->ctx, PFV_ERR_STATE, "an earlier batch failed: the stream is incomplete");
    e->batch[e->cur].order.push_back(GopPacket{3, 0, 0});
    return PFV_OK;
}
// every frame handed over so far becomes packets now (both batches, in stream order)
PFV_API int pfv_gop_encoder_flush(pfv_gop_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->failed) return fail(e->ctx, PFV_ERR_STATE, "an earlier batch failed: the stream is incomplete");
    pfv_ctx *ctx = e->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = gop_enc_submit(e, e->batch[e->cur]);            // its kernels run while the older batch's payloads come over
    if (!rc) rc = gop_enc_collect(e, e->batch[e->cur ^ 1]);  // packets of the older batch first
    if (!rc) rc = gop_enc_collect(e, e->batch[e->cur]);
    if (rc) e->failed = true;
    return rc;
}
// Encoder::finish (src/enc.rs:182-188)
PFV_API int pfv_gop_encoder_finish(pfv_gop_encoder *e)
{
    if (!e) return fail(nullptr, PFV_ERR_BAD_ARG, "null encoder");
    if (e->finished) return fail(e->ctx, PFV_ERR_STATE, "encoder already finished (src/enc.rs:183)");
    int rc = pfv_gop_encoder_flush(e);
    if (rc) return rc;
    e->finished = true;
    static const uint8_t eof[5] = {0, 0, 0, 0, 0};                               // src/enc.rs:221-227
    e->segs.push_back(pfv_iovec{eof, 5});
    return PFV_OK;
}
PFV_API int pfv_gop_encoder_bytes(pfv_gop_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_gop_encoder_bytes: bad argument");
    gop_enc_materialize(e);
    *data = e->out.data();
    *len = e->out.size();
    return PFV_OK;
}
PFV_API int pfv_gop_encoder_drain(pfv_gop_encoder *e, const uint8_t **data, size_t *len)
{
    if (!e || !data || !len) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_gop_encoder_drain: bad argument");
    gop_enc_materialize(e);
    e->drained.swap(e->out);
    e->out.clear();
    *data = e->drained.data();
    *len = e->drained.size();
    return PFV_OK;
}
// The writer side without a copy (the reference's W: Write takes the packets one write_all at a time, src/enc.rs:190-235): the bytes
// produced since the last drain as `count` segments in stream order -- packet headers, and payloads where the device-to-host copy put
// them (page-locked memory).  Valid until the next call on this encoder.
PFV_API int pfv_gop_encoder_drain_iov(pfv_gop_encoder *e, const pfv_iovec **iov, size_t *count)
{
    if (!e || !iov || !count) return fail(e ? e->ctx : nullptr, PFV_ERR_BAD_ARG, "pfv_gop_encoder_drain_iov: bad argument");
    e->segs_drained.clear();
    e->drained.swap(e->out);
    e->out.clear();
    if (!e->drained.empty()) e->segs_drained.push_back(pfv_iovec{e->drained.data(), e->drained.size()});
    e->segs_drained.insert(e->segs_drained.end(), e->segs.begin(), e->segs.end());
    e->segs.clear();
    e->segs_in = 0;
    *iov = e->segs_drained.data();
    *count = e->segs_drained.size();
    return PFV_OK;
}

PFV_API long pfv_gop_encoder_batches(const pfv_gop_encoder *e) { return e ? e->batches : 0; }
/* host seconds so far: out[0] waiting for plane uploads, [1] enqueueing batches, [2] waiting for a batch's kernels, [3] payloads device ->
 * host, [4] packet assembly; returns the number of entries written (<= n) */
PFV_API int pfv_gop_encoder_stats(const pfv_gop_encoder *e, double *out, int n)
{
    if (!e || !out) return 0;
    const int k = std::min(n, 7);
    for (int i = 0; i < k; i++) out[i] = i < 5 ? e->stats[i] : (i == 5 ? (double)e->frames_by_reference : (double)e->batches_redone);
    return k;
}

}  // extern "C"

// ================================================================== GOP-batched decoder
struct GopDecEvent {
    enum Kind { FRAME, DROP, END, ERROR } kind = END;
    int rc = 0;                          // ERROR: status; FRAME: parse / decode status (set while the batch is decoded)
    const char *msg = "";
    uint8_t type = 0;                    // FRAME: 1 / 2
    int slot = 0, t = 0;
    const uint8_t *payload = nullptr;
    uint32_t plen = 0;
    size_t pos_after = 0;
};

struct GopDecSet {   // host staging of one frame step (two alternate: the parse of step t + 1 runs under the device work of step t)
    PinnedBuf<uint32_t> idx;
    PinnedBuf<int16_t> val;
    PinnedBuf<uint32_t> counts;
    PinnedBuf<int8_t> mv;
    PinnedBuf<uint8_t> has;
    PinnedBuf<int> flags;                // bad-motion-vector flags of the step, one per slot
    std::vector<int> rc;                 // per slot: parse status (kSinkFull: dense fallback)
    std::vector<uint8_t> qidx;           // per slot x 3
    std::vector<GopDecEvent *> ev;       // per slot: the packet of this step, or null
    hipEvent_t done = nullptr;           // the device has finished reading this set
    bool used = false;
    int pending = 0;                     // parse tasks of this set not yet finished (under the pool's mutex)
};

// staging sets in rotation: the packets of up to kGopDecSets - 1 frame steps are being parsed while the device works on a step, so the
// parser pool always has a few dozen packets to choose from (one step of a 4K stream is 20 packets of ~6 ms: too few for 16 cores
// to stay busy across the step boundaries)
constexpr int kGopDecSets = 4;

// one packet of a batch on the device-entropy path
struct GopDevPacket {
    GopDecEvent *ev = nullptr;
    int rc = 0;                          // status the frame is delivered with (0: decoded)
    bool host_parse = false;             // the host parser reads it (degenerate table, oversize, or the device stage was not sure)
    uint8_t qidx[3] = {0, 0, 0};
    size_t frame = 0;                    // t * max_gops + slot: its place in the batch-wide arrays
};

constexpr int kGopDevDense = 8;

// device-entropy path of the decoder (PFV_OPT_ENTROPY_DECODE): the whole batch's payloads are read by the k_entd_* kernels
struct GopDecDev {
    bool on = false;
    size_t frames_cap = 0;               // max_gops * max_gop_frames
    uint8_t *bytes_dev = nullptr; size_t bytes_cap = 0;
    EdPacket *pk_dev = nullptr; uint32_t *status_dev = nullptr; size_t pk_cap = 0;
    uint2 *groups_dev = nullptr; size_t groups_cap = 0;
    uint32_t *sub_dev = nullptr; size_t sub_cap = 0;     // end | used | cnt (a fourth of the array each; the last fourth is spare)
    unsigned long long *wgsum_dev = nullptr; size_t wgsum_cap = 0;   // per workgroup
    uint32_t *hdr_maps_dev = nullptr; size_t hdr_maps_cap = 0;       // k_hdr_*: [header workgroup][8]
    uint4 *hdr_start_dev = nullptr; size_t hdr_start_cap = 0;        // [header workgroup]
    uint32_t *coded_dev = nullptr;
    ListPool lists;                      // the batch's coefficients: one list per frame (pfv_device.h: CoefLists), no dense arrays
    std::vector<size_t> list_off, list_room;   // per packet: its list's place in the pool (entd_pool_cap entries)
    int8_t *mv_dev = nullptr;
    uint8_t *has_dev = nullptr;
    // the entropy stage's own streams: it works ahead of the decode kernels and their downloads, and the windows take the streams in turn, so
    // that one window's settling tail (a few lanes in a few wavefronts, round after round) runs beside the next windows' full reads.
    // Measured, config 4 with the frames left in HBM (tools/gpu_inner_sweep.sh, three passes each): one stream 1.25-1.27 G macroblocks/s at
    // the device's greatest stream priority (1.22-1.29 without), two / three streams 1.12-1.24 / 1.16-1.25 (0.89-1.10 without priority: the
    // decode launches then wait behind them), four streams at normal priority 1.27-1.40.  Re-measured at the end of round 5 on one box, passes
    // interleaved (tools/e2e_native.cpp; PFV_GOPD_WINDOW_STREAMS): four 1.14-1.18 G, THREE 1.27-1.34, two 1.03-1.05; the Python probe's wait for
    // the entropy stage 8.0-8.6 / 6.9-7.4 / 9.5-9.7 ms.  (The runtime multiplexes a process's streams over four hardware queues; with the context's
    // stream and the upload stream four window streams make six.  The upload stream at the greatest priority -- a queue pool of its own -- also
    // gave 1.21-1.24 with four; the two together nothing more.)
    static constexpr int kStreams = 4;
    hipStream_t streams[kStreams] = {nullptr, nullptr, nullptr, nullptr};
    int n_streams = 3;
    hipStream_t up_stream = nullptr;     // ... and the uploads / clears it needs run ahead of it on a third
    std::vector<hipEvent_t> window_done; // per step: payloads read, statuses on the host
    std::vector<hipEvent_t> window_up;   // per step: payloads, headers and cleared coefficient arrays in place
    PinnedBuf<uint8_t> bytes_host, has_host;
    PinnedBuf<int8_t> mv_host;
    PinnedBuf<EdPacket> pk_host;
    PinnedBuf<uint2> groups_host;
    uint32_t sub_bits = kEdSubBits;      // payload bits per lane (PFV_OPT_ENTDEC_LANE_BITS)
    int launches = 3, inner = kEdInner;  // read launches before the verifying one (k_entd_sync + launches - 1 x k_entd_fix), rounds inside the first (PFV_OPT_ENTDEC_LAUNCHES / _INNER_ROUNDS)
    long unsettled = 0, irregular = 0;   // why packets were left to the host parser
    PinnedBuf<uint32_t> status_host;
    PinnedBuf<int> flags_host;           // [step][max_gops]
    PinnedBuf<uint32_t> hp_ent;          // up to kGopDevDense packets the host parser reads, in list form: entries (a packet's share: its place's size) ...
    PinnedBuf<uint32_t> hp_counts;       // ... and counts [kGopDevDense][tb + 1]
    PinnedBuf<uint32_t> hp_full;         // one packet whose list outgrew its place (tb x 256 entries)
    size_t hp_off[kGopDevDense] = {}, hp_n[kGopDevDense] = {};
    std::vector<GopDevPacket> pk;
    std::vector<int> todo;               // phase 2: the packets the host parser reads, kGopDevDense at a time
    int todo_first = 0;
    int pending = 0;                     // phase-2 tasks (packets parsed on the host) not yet finished (under the pool's mutex)
    std::vector<int> step_pending;       // phase-1 tasks (headers) not yet finished, per frame step
    long packets_dev = 0, packets_host = 0, batches_dev = 0, batches_host = 0;
};

struct pfv_gop_decoder {
    pfv_ctx *ctx = nullptr;
    pfv_dec_session *hot = nullptr;
    GopDecDev dev;
    const uint8_t *data = nullptr;
    size_t len = 0, pos = 0, reset_pos = 0;
    int width = 0, height = 0, framerate = 0, n_qtables = 0, max_gops = 0, max_len = 0;
    size_t total_blocks = 0, frame_bytes = 0, cap = 0;
    bool eof = false;
    double delta_accum = 0.0;
    // the current batch
    std::vector<GopDecEvent> events;     // stream order
    size_t next_event = 0;
    std::vector<int> glen;               // frames per group
    std::vector<uint8_t> gfirst;         // type of the group's first frame
    bool cont_valid = false;
    int cont_buf = 0, cont_slot = 0;     // where the framebuffer of the previous batch's last group lives (buffer, slot)
    GopDecSet set[kGopDecSets];
    PinnedBuf<int16_t> dense;            // one slot's coefficients when its list overflowed
    PinnedBuf<uint8_t> frames_host;      // [max_gop_frames][max_gops][frame_bytes]: the decoded frames of the batch
    uint8_t *frames_dev = nullptr;       // [max_gops][frame_bytes]
    bool out_dev = false;                // pfv_gop_decoder_set_output_device: frames stay in HBM, the callback gets device pointers
    uint8_t *frames_all_dev = nullptr;   // [steps][max_gops][frame_bytes] then
    size_t frames_all_cap = 0;           // bytes
    long batches = 0, dense_packets = 0;
    // seconds: [0] header scan, [1] waiting for packet parsers (the caller parses too), [2] waiting for the device before a staging set
    // can be reused, [3] enqueueing, [4] waiting for the batch's last frames
    double stats[6] = {0, 0, 0, 0, 0, 0};   // [5]: waiting for the device's entropy stage (device-entropy path)
    // worker pool: the packets of a step are parsed in parallel (one task per slot)
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<std::pair<GopDecSet *, int>> tasks;   // (staging set, slot) packets waiting for a parser
    bool quit = false;
};

static void gopd_parse_one(pfv_gop_decoder *d, GopDecSet *s, int k)
{
    GopDecEvent *e = s->ev[(size_t)k];
    s->counts.data()[k] = 0;
    if (!e) { s->rc[(size_t)k] = 0; return; }
    const size_t tb = d->total_blocks;
    SparseSink sink{s->idx.data() + (size_t)k * d->cap, s->val.data() + (size_t)k * d->cap, d->cap};
    sink.offset = (size_t)k * tb * 256;
    uint8_t *q = &s->qidx[(size_t)k * 3];
    const int rc = e->type == 2 ? parse_pframe_to(e->payload, e->plen, (int)tb, d->n_qtables, s->mv.data() + (size_t)k * tb * 2,
                                                  s->has.data() + (size_t)k * tb, sink, q)
                                : parse_iframe_to(e->payload, e->plen, (int)tb, d->n_qtables, sink, q);
    s->counts.data()[k] = rc == 0 ? (uint32_t)sink.n : 0u;
    s->rc[(size_t)k] = rc;
}
static void gopd_dev_task(pfv_gop_decoder *d, int j);
// a queued task: (staging set, slot) = parse that packet into the set's lists; (null, j >= 0) = the headers of packet j of the device-entropy
// path's batch (phase 1); (null, -1 - j) = the j-th packet of the current group the device stage left to the host parser (phase 2)
static void gopd_run_task(pfv_gop_decoder *d, const std::pair<GopDecSet *, int> &job)
{
    if (job.first) gopd_parse_one(d, job.first, job.second);
    else gopd_dev_task(d, job.second);
}
// a task has been run (or dropped): its counters, under the pool's mutex.  Device-path tasks count per phase and, in phase 1 (headers), per frame
// step as well: the windows of a step are enqueued as soon as ITS packets are ready.  True when some waiter may go on.
static bool gopd_task_done(pfv_gop_decoder *d, const std::pair<GopDecSet *, int> &job)
{
    if (job.first) return --job.first->pending == 0;
    GopDecDev &v = d->dev;
    if (job.second < 0) return --v.pending == 0;        // a packet parsed on the host (phase 2)
    return --v.step_pending[(size_t)v.pk[(size_t)job.second].ev->t] == 0;
}
static void gopd_worker(pfv_gop_decoder *d)
{
    std::unique_lock<std::mutex> lk(d->m);
    for (;;) {
        d->cv_work.wait(lk, [&] { return d->quit || !d->tasks.empty(); });
        if (d->quit) return;
        const auto job = d->tasks.front();
        d->tasks.pop_front();
        lk.unlock();
        gopd_run_task(d, job);
        lk.lock();
        if (gopd_task_done(d, job)) d->cv_done.notify_all();
    }
}
static void gopd_start_parse(pfv_gop_decoder *d, GopDecSet *s, int n_slots)
{
    std::lock_guard<std::mutex> lk(d->m);
    for (int k = 0; k < n_slots; k++) {
        s->counts.data()[k] = 0;
        s->rc[(size_t)k] = 0;
        if (s->ev[(size_t)k]) { d->tasks.emplace_back(s, k); s->pending++; }
    }
    d->cv_work.notify_all();
}
static void gopd_join(pfv_gop_decoder *d, int *pending)
{
    std::unique_lock<std::mutex> lk(d->m);
    while (*pending > 0) {
        if (!d->tasks.empty()) {     // the caller parses too (and is the whole pool when there are no workers): any packet will do
            const auto job = d->tasks.front();
            d->tasks.pop_front();
            lk.unlock();
            gopd_run_task(d, job);
            lk.lock();
            if (gopd_task_done(d, job)) d->cv_done.notify_all();
        } else {
            d->cv_done.wait(lk);
        }
    }
}
static void gopd_join_parse(pfv_gop_decoder *d, GopDecSet *s) { gopd_join(d, &s->pending); }
// nothing of an abandoned batch may stay queued (reset, errors): wait for the parsers to let go of the sets
static void gopd_drain_pool(pfv_gop_decoder *d)
{
    std::unique_lock<std::mutex> lk(d->m);
    for (const auto &job : d->tasks) (void)gopd_task_done(d, job);
    d->tasks.clear();
    d->cv_done.wait(lk, [&] {
        for (const GopDecSet &s : d->set)
            if (s.pending > 0) return false;
        for (int p : d->dev.step_pending)
            if (p > 0) return false;
        return d->dev.pending <= 0;
    });
}

// Walks the packet headers from d->pos exactly as the reference's loop does (src/dec.rs:174-222) and cuts the next batch: up to
// max_gops groups of up to max_gop_frames frame packets, a group per i-frame.
static void gopd_scan_batch(pfv_gop_decoder *d)
{
    d->events.clear(); d->next_event = 0; d->glen.clear(); d->gfirst.clear();
    size_t pos = d->pos;
    auto push = [&](GopDecEvent::Kind kind, size_t pos_after) -> GopDecEvent & {
        d->events.emplace_back();
        GopDecEvent &e = d->events.back();
        e.kind = kind; e.pos_after = pos_after;
        return e;
    };
    for (;;) {
        if (pos + 5 > d->len) {
            GopDecEvent &e = push(GopDecEvent::ERROR, pos);
            e.rc = PFV_ERR_IO; e.msg = "unexpected end of stream in a packet header";
            break;
        }
        const uint8_t type = d->data[pos];
        const uint32_t plen = (uint32_t)d->data[pos + 1] | ((uint32_t)d->data[pos + 2] << 8) | ((uint32_t)d->data[pos + 3] << 16) |
                              ((uint32_t)d->data[pos + 4] << 24);
        if (type == 0) { push(GopDecEvent::END, pos + 5); break; }                  // EOF marker (:183-187)
        if (pos + 5 + (size_t)plen > d->len) {
            GopDecEvent &e = push(GopDecEvent::ERROR, pos + 5);
            e.rc = PFV_ERR_IO; e.msg = "packet payload runs past the end of the stream";
            break;
        }
        const size_t after = pos + 5 + (size_t)plen;
        if (type != 1 && type != 2) { pos = after; continue; }                      // unknown packet: skipped (:216-219)
        if (type == 1 && plen == 0) { push(GopDecEvent::DROP, after); pos = after; continue; }   // drop frame (:190)
        if (type == 1) {
            if ((int)d->glen.size() == d->max_gops) break;                          // the next batch starts here
            d->glen.push_back(0); d->gfirst.push_back(1);
        } else if (d->glen.empty() || d->glen.back() == d->max_len) {
            if (!d->glen.empty()) break;                                            // a run longer than a batch holds: it continues in the next one
            d->glen.push_back(0); d->gfirst.push_back(2);
        }
        GopDecEvent &e = push(GopDecEvent::FRAME, after);
        e.type = type; e.payload = d->data + pos + 5; e.plen = plen;
        e.slot = (int)d->glen.size() - 1; e.t = d->glen.back()++;
        pos = after;
    }
}

// where a batch's frames go: page-locked host memory [step][slot] (the default), or a device array of the same shape
static int gopd_out_room(pfv_gop_decoder *d, int steps)
{
    pfv_ctx *ctx = d->ctx;
    const size_t need = (size_t)std::max(steps, 1) * (size_t)d->max_gops * d->frame_bytes;
    if (!d->out_dev) return d->frames_host.resize(need) ? PFV_OK : fail(ctx, PFV_ERR_NOMEM, "pinned frame staging");
    if (need <= d->frames_all_cap) return PFV_OK;
    if (d->frames_all_dev) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(d->frames_all_dev); d->frames_all_dev = nullptr; d->frames_all_cap = 0; }
    HIP_TRY(ctx, hipMalloc((void **)&d->frames_all_dev, need));
    d->frames_all_cap = need;
    return PFV_OK;
}
// the retframes of step t, slots [first, first + count), after dec_launch: the separate crop pass where the fused one does not apply, and
// the way to the host
static int gopd_step_out(pfv_gop_decoder *d, int t, int first, int count)
{
    pfv_ctx *ctx = d->ctx;
    pfv_dec_session *hot = d->hot;
    int rc = PFV_OK;
    if (!fused_output_ok(hot)) {             // geometries without 16-byte rows: on the buffer just written
        hot->cur ^= 1;
        rc = dec_crop_win(hot, first, count, hot->frames_out, 0);
        hot->cur ^= 1;
    }
    if (!rc && !d->out_dev &&
        hipMemcpyAsync(d->frames_host.data() + ((size_t)t * (size_t)d->max_gops + (size_t)first) * d->frame_bytes, d->frames_dev + (size_t)first * d->frame_bytes,
                       (size_t)count * d->frame_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = fail(ctx, PFV_ERR_HIP, "retframe download");
    return rc;
}
static void gopd_step_target(pfv_gop_decoder *d, int t)
{
    d->hot->frames_out = d->out_dev ? d->frames_all_dev + (size_t)t * (size_t)d->max_gops * d->frame_bytes : d->frames_dev;
}

// decode every frame packet of the scanned batch; the frames land in frames_host[step][slot]
static int gopd_decode_batch(pfv_gop_decoder *d)
{
    pfv_ctx *ctx = d->ctx;
    pfv_dec_session *hot = d->hot;
    const int G = (int)d->glen.size();
    if (G == 0) return PFV_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t tb = d->total_blocks, pad = (size_t)hot->geom.pad_frame_bytes;
    static const char *kBadPayload = "malformed packet payload", *kBadMv = "motion vector points outside the reference plane (src/common.rs:258-259)";

    // chains: the frame packets a slot decodes one after the other.  To begin with, chain k = group k.
    std::vector<std::vector<GopDecEvent *>> chain((size_t)G);
    for (GopDecEvent &e : d->events)
        if (e.kind == GopDecEvent::FRAME) { e.rc = 0; chain[(size_t)e.slot].push_back(&e); }
    auto wait_set = [&](GopDecSet &s) -> int {   // the device may still be reading the set's lists; then attribute the step's flags
        if (!s.used) return PFV_OK;
        GopClock wclk;
        HIP_TRY(ctx, hipEventSynchronize(s.done));
        d->stats[2] += wclk.lap();
        for (int k = 0; k < G; k++)
            if (s.flags.data()[k] && s.ev[(size_t)k] && !s.ev[(size_t)k]->rc) { s.ev[(size_t)k]->rc = PFV_ERR_BAD_MV; s.ev[(size_t)k]->msg = kBadMv; }
        s.used = false;
        return PFV_OK;
    };
    auto fill = [&](GopDecSet &s, int t) {
        for (int k = 0; k < G; k++) s.ev[(size_t)k] = (int)chain[(size_t)k].size() > t ? chain[(size_t)k][(size_t)t] : nullptr;
    };
    // status of a parsed packet: 0 (kSinkFull counts: it is parsed again into the dense form) or the error it is delivered with
    auto status = [&](const GopDecSet &s, int k) -> int {
        int prc = s.rc[(size_t)k];
        if (prc == kSinkFull) prc = 0;
        const uint8_t *q = &s.qidx[(size_t)k * 3];
        if (!prc)
            for (int i = 0; i < 3; i++)
                if (q[i] >= hot->n_qtables) prc = PFV_ERR_FORMAT;              // the reference panics (src/dec.rs:249-251)
        return prc;
    };
    int rc = PFV_OK;
    for (GopDecSet &s : d->set)
        if (!rc) rc = wait_set(s);
    if (rc) return rc;

    // step 0 is parsed before anything runs: a group whose i-frame does not parse is no independent run -- the sequential loop
    // leaves the framebuffer alone and applies the group's p-frames to what the PREVIOUS group left (src/dec.rs:188-214).  Such a
    // group is appended to the chain of the slot before it (slot 0: it continues the run of the previous batch).
    GopClock clk;
    fill(d->set[0], 0);
    gopd_start_parse(d, &d->set[0], G);
    // The steps behind it go to the parsers at the same time, on the assumption that every group's first frame is sound (the i-frames of
    // step 0 are several times the size of a p-frame: without this the pool idles while the slowest of them is read).  If one is not,
    // the chains change and these steps are parsed again.
    int prefilled = 1;
    {
        int steps0 = 0;
        for (int k = 0; k < G; k++) steps0 = std::max(steps0, (int)chain[(size_t)k].size());
        for (int t = 1; t < kGopDecSets && t < steps0; t++) {
            fill(d->set[t], t);
            gopd_start_parse(d, &d->set[t], G);
            prefilled = t + 1;
        }
    }
    gopd_join_parse(d, &d->set[0]);
    // An i-frame whose list overflowed (denser than 1 non-zero in 4) was not read to its end: whether it parses is only known after a
    // full pass, and the chains below depend on it -- read it once more with a sink that keeps nothing (dense i-frames only: rare)
    for (int k = 0; k < G; k++) {
        GopDecEvent *e0 = d->set[0].ev[(size_t)k];
        if (!e0 || e0->type != 1 || d->set[0].rc[(size_t)k] != kSinkFull) continue;
        struct { bool put(size_t, int16_t) { return true; } bool put_if(bool, size_t, int16_t) { return true; } } none;
        uint8_t q[3];
        const int vrc = parse_iframe_to(e0->payload, e0->plen, (int)tb, d->n_qtables, none, q);
        if (vrc) d->set[0].rc[(size_t)k] = vrc;
    }
    d->stats[1] += clk.lap();
    bool reparse = false, head_continues = d->gfirst[0] == 2;
    int last_root = G - 1;
    {
        std::vector<int> root((size_t)G);
        for (int k = 0; k < G; k++) {
            root[(size_t)k] = k;
            GopDecEvent *e0 = chain[(size_t)k].empty() ? nullptr : chain[(size_t)k][0];
            if (!e0 || e0->type != 1) continue;
            const int prc = status(d->set[0], k);
            if (!prc) continue;
            e0->rc = prc; e0->msg = kBadPayload;
            reparse = true;
            const int r = k == 0 ? 0 : root[(size_t)k - 1];
            root[(size_t)k] = r;
            std::vector<GopDecEvent *> rest(chain[(size_t)k].begin() + 1, chain[(size_t)k].end());
            if (k == 0) { chain[0] = rest; head_continues = true; }
            else { chain[(size_t)k].clear(); chain[(size_t)r].insert(chain[(size_t)r].end(), rest.begin(), rest.end()); }
        }
        last_root = root[(size_t)G - 1];
    }
    int steps = 0;
    for (int k = 0; k < G; k++) {
        steps = std::max(steps, (int)chain[(size_t)k].size());
        for (size_t t = 0; t < chain[(size_t)k].size(); t++) { chain[(size_t)k][t]->slot = k; chain[(size_t)k][t]->t = (int)t; }
    }
    if (head_continues && d->cont_valid) {   // slot 0 continues the run the previous batch left open
        const uint8_t *src = hot->fb[d->cont_buf] + (size_t)d->cont_slot * pad;
        uint8_t *dst = hot->fb[hot->cur];
        if (src != dst) HIP_TRY(ctx, hipMemcpyAsync(dst, src, pad, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // merged chains may be longer than a group
    if ((rc = gopd_out_room(d, steps))) return rc;
    const int cur0 = hot->cur;
    // steps [t, next_fill) are parsed or being parsed; step s uses staging set s % kGopDecSets.  A set is refilled as soon as the device
    // has finished with the step that used it last.
    if (reparse) gopd_drain_pool(d);         // the chains moved: what was parsed ahead belongs to other (slot, step) places now
    int next_fill = reparse ? 0 : prefilled;
    auto top_up = [&](int t, bool must_have_t) -> int {
        while (next_fill < steps && next_fill < t + kGopDecSets) {
            GopDecSet &n = d->set[next_fill % kGopDecSets];
            if (n.used && !(must_have_t && next_fill <= t) && hipEventQuery(n.done) != hipSuccess) { (void)hipGetLastError(); break; }
            const int wrc = wait_set(n);
            if (wrc) return wrc;
            fill(n, next_fill);
            gopd_start_parse(d, &n, G);
            next_fill++;
        }
        return PFV_OK;
    };
    std::vector<int> key((size_t)G);
    std::vector<uint32_t> combos;            // distinct (frame type, q-table indices) of a step -> launch key
    for (int t = 0; t < steps; t++) {
        GopDecSet &s = d->set[t % kGopDecSets];
        clk.lap();
        if ((rc = top_up(t, true))) return rc;
        d->stats[3] += clk.lap();
        gopd_join_parse(d, &s);
        d->stats[1] += clk.lap();
        // what runs: the packets that parsed.  A failed packet changes nothing (its error surfaces when the frame is delivered), but its
        // slot's framebuffer has to follow the ping-pong for the frames behind it.
        bool any_dense = false;
        combos.clear();
        for (int k = 0; k < G; k++) {
            GopDecEvent *e = s.ev[(size_t)k];
            key[(size_t)k] = -1;
            if (!e) continue;
            const int prc = status(s, k);
            if (prc) {
                e->rc = prc; e->msg = kBadPayload;
                HIP_TRY(ctx, hipMemcpyAsync(hot->fb[hot->cur ^ 1] + (size_t)k * pad, hot->fb[hot->cur] + (size_t)k * pad, pad, hipMemcpyDeviceToDevice, ctx->stream));
                continue;
            }
            any_dense = any_dense || s.rc[(size_t)k] == kSinkFull;
            const uint8_t *q = &s.qidx[(size_t)k * 3];
            const uint32_t c = (uint32_t)e->type | ((uint32_t)q[0] << 8) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 24);
            size_t ci = 0;
            while (ci < combos.size() && combos[ci] != c) ci++;
            if (ci == combos.size()) combos.push_back(c);
            key[(size_t)k] = (int)ci;
        }
        const size_t total = tb * (size_t)G * 256;
        HIP_TRY(ctx, hipMemsetAsync(hot->st_coef, 0, total * 2, ctx->stream));
        hipLaunchKernelGGL(k_scatter_coef_seg, dim3(64, (unsigned)G), dim3(kThreads), 0, ctx->stream, s.idx.data(), s.val.data(), s.counts.data(),
                           (uint32_t)d->cap, (uint32_t)total, hot->st_coef);
        if ((rc = launch_check(ctx, "k_scatter_coef_seg"))) return rc;
        if (any_dense) {   // a list overflowed (denser than 1 non-zero in 4): that packet again, into the dense form, on this thread
            for (int k = 0; k < G; k++) {
                GopDecEvent *e = s.ev[(size_t)k];
                if (!e || s.rc[(size_t)k] != kSinkFull || key[(size_t)k] < 0) continue;
                d->dense_packets++;
                if (!d->dense.resize(tb * 256)) return fail(ctx, PFV_ERR_NOMEM, "pinned dense staging");
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));        // the previous user of the dense buffer
                memset(d->dense.data(), 0, tb * 512);
                DenseSink sink{d->dense.data()};
                uint8_t q[3];
                const int prc = e->type == 2 ? parse_pframe_to(e->payload, e->plen, (int)tb, d->n_qtables, s.mv.data() + (size_t)k * tb * 2,
                                                               s.has.data() + (size_t)k * tb, sink, q)
                                             : parse_iframe_to(e->payload, e->plen, (int)tb, d->n_qtables, sink, q);
                if (prc) {
                    e->rc = prc; e->msg = kBadPayload; key[(size_t)k] = -1;
                    HIP_TRY(ctx, hipMemcpyAsync(hot->fb[hot->cur ^ 1] + (size_t)k * pad, hot->fb[hot->cur] + (size_t)k * pad, pad, hipMemcpyDeviceToDevice, ctx->stream));
                    continue;
                }
                HIP_TRY(ctx, hipMemcpyAsync(hot->st_coef + (size_t)k * tb * 256, d->dense.data(), tb * 512, hipMemcpyHostToDevice, ctx->stream));
            }
        }
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_mv, s.mv.data(), (size_t)G * tb * 2, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(hot->st_has, s.has.data(), (size_t)G * tb, hipMemcpyHostToDevice, ctx->stream));
        rc = PFV_OK;
        gopd_step_target(d, t);
        gop_runs(key, [&](int first, int count, int ci) {
            if (rc) return;
            const uint32_t c = combos[(size_t)ci];
            const uint8_t q[3] = {(uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
            rc = dec_launch(hot, (c & 0xffu) == 2, first, count, hot->st_mv, hot->st_has, hot->st_coef, q);
            if (!rc) rc = gopd_step_out(d, t, first, count);
        });
        if (rc) return rc;
        hot->cur ^= 1;
        HIP_TRY(ctx, hipMemcpyAsync(s.flags.data(), hot->flag_dev, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(hot->flag_dev, 0, (size_t)G * sizeof(int), ctx->stream));
        HIP_TRY(ctx, hipEventRecord(s.done, ctx->stream));
        s.used = true;
        if ((rc = top_up(t + 1, false))) return rc;
        d->stats[3] += clk.lap();
    }
    clk.lap();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    d->stats[4] += clk.lap();
    for (GopDecSet &s : d->set)
        if ((rc = wait_set(s))) return rc;
    // the run of the last group may go on in the next batch: its framebuffer is in the buffer its chain's last step wrote
    d->cont_valid = true;
    d->cont_slot = last_root;
    d->cont_buf = (cur0 + (int)chain[(size_t)last_root].size()) & 1;
    d->batches++;
    return PFV_OK;
}

// ---------------------------------------------------------------- device-entropy path (PFV_OPT_ENTROPY_DECODE)
// phase 1, per packet: what the kernels need that only a serial read can give -- the table (-> the tree's codes), the q indices, a
// p-frame's block headers (-> motion vectors, has_coeff, the first bit of the run streams); the payload goes to page-locked staging
static void gopd_dev_prepare(pfv_gop_decoder *d, int j)
{
    GopDecDev &v = d->dev;
    GopDevPacket &p = v.pk[(size_t)j];
    const GopDecEvent *e = p.ev;
    EdPacket &k = v.pk_host.data()[j];
    const size_t tb = d->total_blocks;
    const EntdPrep r = entd_prepare(e->payload, e->plen, e->type, tb, d->n_qtables, v.sub_bits, k, v.bytes_host.data() + k.byte_off);
    p.rc = r.rc;
    p.host_parse = r.host_parse;
    memcpy(p.qidx, r.qidx, 3);
}
// phase 2, per packet the device stage left to the host: the host parser, into a dense frame
static void gopd_dev_hostparse(pfv_gop_decoder *d, int j)
{
    GopDecDev &v = d->dev;
    GopDevPacket &p = v.pk[(size_t)v.todo[(size_t)(v.todo_first + j)]];
    const GopDecEvent *e = p.ev;
    const size_t tb = d->total_blocks, pj = (size_t)v.todo[(size_t)(v.todo_first + j)];
    uint8_t q[3];
    p.rc = parse_to_lists(e->payload, e->plen, e->type, tb, d->n_qtables, v.mv_host.data() + p.frame * tb * 2, v.has_host.data() + p.frame * tb, v.hp_ent.data() + v.hp_off[j],
                          v.list_room[pj], v.hp_counts.data() + (size_t)j * (tb + 1), &v.hp_n[j], q);
}
static void gopd_dev_task(pfv_gop_decoder *d, int code)
{
    if (code >= 0) gopd_dev_prepare(d, code);
    else gopd_dev_hostparse(d, -1 - code);
}
// phase 2: n_tasks packets (v.todo from todo_first on) through the host parser, on the pool and this thread
static void gopd_dev_hostparse_group(pfv_gop_decoder *d, int n_tasks)
{
    {
        std::lock_guard<std::mutex> lk(d->m);
        for (int j = 0; j < n_tasks; j++) { d->tasks.emplace_front(nullptr, -1 - j); d->dev.pending++; }   // ahead of the headers still queued: a step is waiting
        d->cv_work.notify_all();
    }
    gopd_join(d, &d->dev.pending);
}
// a p-frame packet the HOST parser read: its block headers go up with its lists (the device's own read of them is not what is decoded)
static int gopd_dev_upload_headers(pfv_gop_decoder *d, const GopDevPacket &p)
{
    pfv_ctx *ctx = d->ctx;
    GopDecDev &v = d->dev;
    const size_t tb = d->total_blocks;
    if (p.ev->type != 2) return PFV_OK;
    HIP_TRY(ctx, hipMemcpyAsync(v.mv_dev + p.frame * tb * 2, v.mv_host.data() + p.frame * tb * 2, tb * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(v.has_dev + p.frame * tb, v.has_host.data() + p.frame * tb, tb, hipMemcpyHostToDevice, ctx->stream));
    return PFV_OK;
}
template <class T>
static int gopd_dev_room(pfv_ctx *ctx, T **p, size_t *cap, size_t need)
{
    if (need <= *cap) return PFV_OK;
    if (*p) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    need += need / 4;
    HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(T)));
    *cap = need;
    return PFV_OK;
}

// Decode the scanned batch with the payloads read on the device.  PFV_OK: done (frames in frames_host[step][slot]); 1: this batch needs
// the host path (a group's i-frame does not parse: the chains change, see gopd_decode_batch) -- nothing has been decoded; negative: error.
//
// The entropy stage of step t (payload upload, k_entd_*, status download) runs on a stream of its own, one window per step, all windows
// enqueued up front: while the context's stream decodes step t and sends its frames to the host -- the PCIe time that bounds the whole
// decoder -- the device reads the payloads of the steps behind it.
static int gopd_decode_batch_dev(pfv_gop_decoder *d)
{
    pfv_ctx *ctx = d->ctx;
    pfv_dec_session *hot = d->hot;
    GopDecDev &v = d->dev;
    const int G = (int)d->glen.size();
    if (G == 0) return PFV_OK;
    int steps = 0;
    for (int k = 0; k < G; k++) steps = std::max(steps, d->glen[(size_t)k]);
    if (steps > d->max_len || G > d->max_gops) return 1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t tb = d->total_blocks, pad = (size_t)hot->geom.pad_frame_bytes, S = (size_t)d->max_gops;
    static const char *kBadPayload = "malformed packet payload", *kBadMv = "motion vector points outside the reference plane (src/common.rs:258-259)";
    GopClock clk;
    HIP_TRY(ctx, hipStreamSynchronize(v.up_stream));       // a batch that went to the host path may have left windows behind
    for (int k = 0; k < v.n_streams; k++) HIP_TRY(ctx, hipStreamSynchronize(v.streams[k]));

    // packets in (step, slot) order: a step's packets, payload bytes, subsequences and workgroups are contiguous
    v.pk.clear();
    for (GopDecEvent &e : d->events)
        if (e.kind == GopDecEvent::FRAME) {
            GopDevPacket p;
            p.ev = &e;
            p.frame = (size_t)e.t * S + (size_t)e.slot;
            v.pk.push_back(p);
        }
    std::sort(v.pk.begin(), v.pk.end(), [](const GopDevPacket &a, const GopDevPacket &b) { return a.frame < b.frame; });
    const size_t n = v.pk.size();
    if (!v.pk_host.resize(n) || !v.status_host.resize(n)) return fail(ctx, PFV_ERR_NOMEM, "device-entropy staging");
    std::vector<size_t> p0((size_t)steps + 1, n), byte0((size_t)steps + 1, 0);
    size_t bytes_total = 0;
    for (size_t j = 0; j < n; j++) {
        EdPacket &k = v.pk_host.data()[j];
        const size_t t = (size_t)v.pk[j].ev->t;
        if (p0[t] == n) { p0[t] = j; byte0[t] = bytes_total; }
        k.byte_off = bytes_total;
        k.frame_off = v.pk[j].frame;
        bytes_total += ((size_t)v.pk[j].ev->plen + 16 + 15) & ~(size_t)15;
    }
    byte0[(size_t)steps] = bytes_total;
    for (size_t t = (size_t)steps; t-- > 0;)
        if (p0[t] == n) { p0[t] = p0[t + 1]; byte0[t] = byte0[t + 1]; }
    if (!v.bytes_host.resize(bytes_total + 64)) return fail(ctx, PFV_ERR_NOMEM, "device-entropy payload staging");
    // room on the device from upper bounds (a lane per sub_bits payload bits), so that nothing has to wait for the headers
    size_t sub_max = 0, grp_max = 0;
    for (size_t j = 0; j < n; j++) {
        const size_t lanes = ((size_t)v.pk[j].ev->plen * 8 + v.sub_bits - 1) / v.sub_bits;
        sub_max += lanes;
        grp_max += (lanes + kEdOwn - 1) / kEdOwn;
    }
    if (sub_max >= 0xffffffffull) return 1;
    int rc = PFV_OK;
    if (!v.groups_host.resize(std::max<size_t>(grp_max, 1))) return fail(ctx, PFV_ERR_NOMEM, "device-entropy staging");
    if ((rc = gopd_dev_room(ctx, &v.bytes_dev, &v.bytes_cap, bytes_total + 64))) return rc;
    if (n > v.pk_cap) {
        if (v.pk_dev) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(v.pk_dev); (void)hipFree(v.status_dev); v.pk_dev = nullptr; v.status_dev = nullptr; v.pk_cap = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&v.pk_dev, (n + n / 4) * sizeof(EdPacket)));
        HIP_TRY(ctx, hipMalloc((void **)&v.status_dev, (n + n / 4) * sizeof(uint32_t)));
        v.pk_cap = n + n / 4;
    }
    if ((rc = gopd_dev_room(ctx, &v.groups_dev, &v.groups_cap, std::max<size_t>(grp_max, 1)))) return rc;
    if ((rc = gopd_dev_room(ctx, &v.sub_dev, &v.sub_cap, std::max<size_t>(sub_max, 1) * 4))) return rc;
    if ((rc = gopd_dev_room(ctx, &v.wgsum_dev, &v.wgsum_cap, std::max<size_t>(grp_max, 1)))) return rc;
    size_t hdr_max = 0;
    for (size_t j = 0; j < n; j++) hdr_max += v.pk[j].ev->type == 2 ? entd_hdr_wgs(tb, v.pk[j].ev->plen) : 0;
    if ((rc = gopd_dev_room(ctx, &v.hdr_maps_dev, &v.hdr_maps_cap, (hdr_max + 1) * 8))) return rc;
    if ((rc = gopd_dev_room(ctx, &v.hdr_start_dev, &v.hdr_start_cap, hdr_max + 1))) return rc;
    // the coefficient lists: every packet's place in the pool from its size alone (entd_pool_cap), the frames' list pointers with them
    v.list_off.assign(n, 0); v.list_room.assign(n, 0);
    size_t list_total = 0;
    for (size_t j = 0; j < n; j++) {
        v.list_off[j] = list_total;
        v.list_room[j] = entd_pool_cap(tb, v.pk[j].ev->plen);
        list_total += v.list_room[j];
    }
    if (list_total > v.lists.ent_cap) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the previous batch's decode launches read the pool
    v.lists.drop_spill();                                                                     // (ctx->stream is idle between batches: the frames were waited for)
    if ((rc = v.lists.room(ctx, std::max<size_t>(list_total, 4)))) return rc;
    for (size_t f = 0; f < v.frames_cap; f++) v.lists.ptr_host.data()[f] = nullptr;
    for (size_t j = 0; j < n; j++) v.lists.ptr_host.data()[v.pk[j].frame] = v.lists.ent + v.list_off[j];
    HIP_TRY(ctx, hipMemcpyAsync(v.lists.ptr_dev, v.lists.ptr_host.data(), v.frames_cap * sizeof(uint32_t *), hipMemcpyHostToDevice, v.up_stream));
    while (v.window_done.size() < (size_t)steps) {
        hipEvent_t ev = nullptr, ev2 = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        v.window_done.push_back(ev);
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
        v.window_up.push_back(ev2);
    }
    HIP_TRY(ctx, hipMemsetAsync(v.status_dev, 0, n * sizeof(uint32_t), v.up_stream));
    // the packets' headers: on the pool, in (step, slot) order; nobody waits for all of them -- a step's window starts when ITS packets are ready
    {
        std::lock_guard<std::mutex> lk(d->m);
        v.step_pending.assign((size_t)steps, 0);
        for (size_t j = 0; j < n; j++) {
            d->tasks.emplace_back(nullptr, (int)j);
            v.step_pending[(size_t)v.pk[j].ev->t]++;
        }
        d->cv_work.notify_all();
    }
    const size_t ts = v.sub_cap / 4;
    size_t total_sub = 0, n_groups = 0, hdr_total = 0;
    int next_window = 0;
    // the windows of steps [next_window, upto]: uploads and clears on one stream, the kernels behind them on another
    auto windows_upto = [&](int upto) -> int {
        for (; next_window <= upto && next_window < steps; next_window++) {
            const int t = next_window;
            GopClock wclk;
            gopd_join(d, &v.step_pending[(size_t)t]);
            d->stats[1] += wclk.lap();
            const size_t f0 = (size_t)t * S, pa = p0[(size_t)t], pb = p0[(size_t)t + 1], ga = n_groups;
            unsigned max_hdr = 0;
            for (size_t j = pa; j < pb; j++) {
                EdPacket &k = v.pk_host.data()[j];
                if (v.pk[j].rc || v.pk[j].host_parse) k.n_sub = k.hdr_wgs = 0;
                k.sub_first = (uint32_t)total_sub;
                k.grp_first = (uint32_t)n_groups;
                k.hdr_first = (uint32_t)hdr_total;
                hdr_total += k.hdr_wgs;
                max_hdr = std::max(max_hdr, (unsigned)k.hdr_wgs);
                total_sub += k.n_sub;
                for (uint32_t b = 0; b * (uint32_t)kEdOwn < k.n_sub; b++) v.groups_host.data()[n_groups++] = make_uint2((unsigned)j, b);
            }
            const size_t gb = n_groups, ba = byte0[(size_t)t], bb = byte0[(size_t)t + 1];
            if (pb > pa) HIP_TRY(ctx, hipMemcpyAsync(v.pk_dev + pa, v.pk_host.data() + pa, (pb - pa) * sizeof(EdPacket), hipMemcpyHostToDevice, v.up_stream));
            if (gb > ga) HIP_TRY(ctx, hipMemcpyAsync(v.groups_dev + ga, v.groups_host.data() + ga, (gb - ga) * sizeof(uint2), hipMemcpyHostToDevice, v.up_stream));
            if (bb > ba) HIP_TRY(ctx, hipMemcpyAsync(v.bytes_dev + ba, v.bytes_host.data() + ba, bb - ba, hipMemcpyHostToDevice, v.up_stream));
            HIP_TRY(ctx, hipEventRecord(v.window_up[(size_t)t], v.up_stream));
            const hipStream_t es = v.streams[t % v.n_streams];
            HIP_TRY(ctx, hipStreamWaitEvent(es, v.window_up[(size_t)t], 0));
            if (gb > ga) {
                EdBufs b{v.bytes_dev, v.pk_dev, v.groups_dev + ga, v.sub_dev, v.sub_dev + ts, v.sub_dev + 2 * ts, v.wgsum_dev, v.coded_dev, v.lists.ptr_dev, v.lists.counts_dev,
                         v.status_dev, (uint32_t)pa, (uint32_t)ga, v.hdr_maps_dev, v.hdr_start_dev, v.mv_dev, v.has_dev};
                entd_launch(es, b, (unsigned)(pb - pa), (unsigned)(gb - ga), max_hdr, v.launches, v.inner);
                const int lrc = launch_check(ctx, "k_entd_*");
                if (lrc) return lrc;
            }
            if (pb > pa) HIP_TRY(ctx, hipMemcpyAsync(v.status_host.data() + pa, v.status_dev + pa, (pb - pa) * sizeof(uint32_t), hipMemcpyDeviceToHost, es));
            HIP_TRY(ctx, hipEventRecord(v.window_done[(size_t)t], es));
            d->stats[3] += wclk.lap();
        }
        return PFV_OK;
    };
    constexpr int kWindowsAhead = 4;     // windows enqueued ahead of the step being decoded (2 .. 15 measured: 1.27-1.38 G whichever, config 4 to HBM)
    if ((rc = windows_upto(kWindowsAhead))) return rc;
    d->stats[3] += clk.lap();

    if (d->gfirst[0] == 2 && d->cont_valid) {   // slot 0 continues the run the previous batch left open
        const uint8_t *src = hot->fb[d->cont_buf] + (size_t)d->cont_slot * pad;
        uint8_t *dst = hot->fb[hot->cur];
        if (src != dst) HIP_TRY(ctx, hipMemcpyAsync(dst, src, pad, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if ((rc = gopd_out_room(d, steps))) return rc;
    if (!v.flags_host.resize((size_t)steps * S)) return fail(ctx, PFV_ERR_NOMEM, "pinned flag staging");
    memset(v.flags_host.data(), 0, (size_t)steps * S * sizeof(int));
    const int cur0 = hot->cur;
    std::vector<int> key((size_t)G);
    std::vector<uint32_t> combos;
    for (int t = 0; t < steps; t++) {
        const size_t f0 = (size_t)t * S, pa = p0[(size_t)t], pb = p0[(size_t)t + 1];
        if ((rc = windows_upto(t + kWindowsAhead))) return rc;
        clk.lap();
        HIP_TRY(ctx, hipEventSynchronize(v.window_done[(size_t)t]));
        d->stats[5] += clk.lap();
        // what the device stage was not sure about goes through the host parser, which decides
        v.todo.clear();
        for (size_t j = pa; j < pb; j++) {
            GopDevPacket &p = v.pk[j];
            if (p.rc) continue;
            const uint32_t st = v.status_host.data()[j];
            if (st & kEdUnsettled) v.unsettled++;
            else if (st) v.irregular++;
            if (p.host_parse || st) { p.host_parse = true; v.todo.push_back((int)j); }
        }
        for (size_t first = 0; first < v.todo.size(); first += (size_t)kGopDevDense) {
            const int cnt = (int)std::min<size_t>((size_t)kGopDevDense, v.todo.size() - first);
            size_t need = 0;
            for (int j = 0; j < cnt; j++) { v.hp_off[j] = need; need += v.list_room[(size_t)v.todo[first + (size_t)j]]; }
            if (!v.hp_ent.resize(need) || !v.hp_counts.resize((size_t)kGopDevDense * (tb + 1))) return fail(ctx, PFV_ERR_NOMEM, "pinned list staging");
            v.todo_first = (int)first;
            gopd_dev_hostparse_group(d, cnt);
            bool overflowed = false;
            for (int j = 0; j < cnt; j++) {
                const size_t pj = (size_t)v.todo[first + (size_t)j];
                const GopDevPacket &p = v.pk[pj];
                if (p.rc == kSinkFull) overflowed = true;
                if (p.rc) continue;
                if ((rc = upload_lists(ctx, v.lists, p.frame, v.list_room[pj], v.hp_ent.data() + v.hp_off[j], v.hp_n[j], v.hp_counts.data() + (size_t)j * (tb + 1), ctx->stream))) return rc;
                if ((rc = gopd_dev_upload_headers(d, p))) return rc;
            }
            // more values than the packet's bits could hold at three bits each (a one-symbol table: values of one or two bits): once more, with
            // room for every coefficient; its list gets a buffer of its own
            for (int j = 0; overflowed && j < cnt; j++) {
                const size_t pj = (size_t)v.todo[first + (size_t)j];
                GopDevPacket &p = v.pk[pj];
                if (p.rc != kSinkFull) continue;
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (!v.hp_full.resize(tb * 256)) return fail(ctx, PFV_ERR_NOMEM, "pinned list staging");
                uint8_t q[3];
                p.rc = parse_to_lists(p.ev->payload, p.ev->plen, p.ev->type, tb, d->n_qtables, v.mv_host.data() + p.frame * tb * 2, v.has_host.data() + p.frame * tb, v.hp_full.data(), tb * 256,
                                      v.hp_counts.data() + (size_t)j * (tb + 1), &v.hp_n[j], q);
                if (!p.rc && (rc = upload_lists(ctx, v.lists, p.frame, v.list_room[pj], v.hp_full.data(), v.hp_n[j], v.hp_counts.data() + (size_t)j * (tb + 1), ctx->stream))) return rc;
                if (!p.rc && (rc = gopd_dev_upload_headers(d, p))) return rc;
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // the staging is used again
        }
        if (!v.todo.empty()) d->stats[1] += clk.lap();
        combos.clear();
        for (int k = 0; k < G; k++) key[(size_t)k] = -1;
        for (size_t j = pa; j < pb; j++) {
            GopDevPacket &p = v.pk[j];
            if (!p.rc)
                for (int i = 0; i < 3; i++)
                    if (p.qidx[i] >= hot->n_qtables) p.rc = PFV_ERR_FORMAT;      // the reference panics (src/dec.rs:249-251)
            if (p.rc && t == 0 && p.ev->type == 1) {                            // not an independent run after all: the chains change
                gopd_drain_pool(d);                                              // (headers still queued or being read belong to this attempt)
                return 1;
            }
        }
        for (size_t j = pa; j < pb; j++) {
            GopDevPacket &p = v.pk[j];
            const int k = p.ev->slot;
            p.ev->rc = p.rc;
            if (p.rc) {   // a failed packet changes nothing, but its slot's framebuffer has to follow the ping-pong
                p.ev->msg = kBadPayload;
                HIP_TRY(ctx, hipMemcpyAsync(hot->fb[hot->cur ^ 1] + (size_t)k * pad, hot->fb[hot->cur] + (size_t)k * pad, pad, hipMemcpyDeviceToDevice, ctx->stream));
                continue;
            }
            if (p.host_parse) v.packets_host++;
            else v.packets_dev++;
            const uint32_t c = (uint32_t)p.ev->type | ((uint32_t)p.qidx[0] << 8) | ((uint32_t)p.qidx[1] << 16) | ((uint32_t)p.qidx[2] << 24);
            size_t ci = 0;
            while (ci < combos.size() && combos[ci] != c) ci++;
            if (ci == combos.size()) combos.push_back(c);
            key[(size_t)k] = (int)ci;
        }
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, v.window_done[(size_t)t], 0));
        rc = PFV_OK;
        gopd_step_target(d, t);
        gop_runs(key, [&](int first, int count, int ci) {
            if (rc) return;
            const uint32_t c = combos[(size_t)ci];
            const uint8_t q[3] = {(uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
            rc = dec_launch(hot, (c & 0xffu) == 2, first, count, v.mv_dev + f0 * tb * 2, v.has_dev + f0 * tb, v.lists.coefs(f0), q);
            if (!rc) rc = gopd_step_out(d, t, first, count);
        });
        if (rc) return rc;
        hot->cur ^= 1;
        HIP_TRY(ctx, hipMemcpyAsync(v.flags_host.data() + f0, hot->flag_dev, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(hot->flag_dev, 0, (size_t)G * sizeof(int), ctx->stream));
        d->stats[3] += clk.lap();
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    d->stats[4] += clk.lap();
    for (size_t j = 0; j < n; j++) {
        GopDecEvent *e = v.pk[j].ev;
        if (v.flags_host.data()[(size_t)e->t * S + (size_t)e->slot] && !e->rc) { e->rc = PFV_ERR_BAD_MV; e->msg = kBadMv; }
    }
    d->cont_valid = true;
    d->cont_slot = G - 1;
    d->cont_buf = (cur0 + d->glen[(size_t)G - 1]) & 1;
    d->batches++;
    v.batches_dev++;
    return PFV_OK;
}

extern "C" {

PFV_API void pfv_gop_decoder_destroy(pfv_gop_decoder *d)
{
    if (!d) return;
    {
        std::lock_guard<std::mutex> lk(d->m);
        d->quit = true;
        d->cv_work.notify_all();
    }
    for (auto &t : d->workers) t.join();
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    for (GopDecSet &s : d->set)
        if (s.done) (void)hipEventDestroy(s.done);
    if (d->frames_dev) (void)hipFree(d->frames_dev);
    if (d->frames_all_dev) (void)hipFree(d->frames_all_dev);
    for (hipStream_t st : d->dev.streams)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (d->dev.up_stream) { (void)hipStreamSynchronize(d->dev.up_stream); (void)hipStreamDestroy(d->dev.up_stream); }
    for (hipEvent_t ev : d->dev.window_done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : d->dev.window_up) (void)hipEventDestroy(ev);
    for (void *p : {(void *)d->dev.bytes_dev, (void *)d->dev.pk_dev, (void *)d->dev.status_dev, (void *)d->dev.groups_dev, (void *)d->dev.sub_dev, (void *)d->dev.coded_dev,
                    (void *)d->dev.wgsum_dev, (void *)d->dev.mv_dev, (void *)d->dev.has_dev, (void *)d->dev.hdr_maps_dev, (void *)d->dev.hdr_start_dev})
        if (p) (void)hipFree(p);
    d->dev.lists.destroy();
    pfv_dec_session_destroy(d->hot);
    delete d;
}

// Decoder::new (src/dec.rs:38-134) + the batch shape (see pfv_gop_encoder_create); n_threads: packet parsers besides the calling thread.
PFV_API int pfv_gop_decoder_create(pfv_ctx *ctx, const uint8_t *data, size_t len, int max_gops, int max_gop_frames, int n_threads,
                                   pfv_gop_decoder **out)
{
    if (!ctx || !data || !out) return fail(ctx, PFV_ERR_BAD_ARG, "pfv_gop_decoder_create: bad argument");
    *out = nullptr;
    if (max_gops <= 0 || max_gop_frames <= 0 || max_gops > 4096 || max_gop_frames > 4096 || n_threads < 0 || n_threads > 256)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_gop_decoder_create: max_gops and max_gop_frames must be in 1..4096, n_threads in 0..256");
    static const char magic[8] = {'P', 'F', 'V', 'I', 'D', 'E', 'O', 0};
    if (len < 8) return fail(ctx, PFV_ERR_IO, "stream shorter than the magic (DecodeError::IOError)");
    if (memcmp(data, magic, 8) != 0) return fail(ctx, PFV_ERR_FORMAT, "bad magic (DecodeError::FormatError, src/dec.rs:50-52)");
    if (len < 12) return fail(ctx, PFV_ERR_IO, "truncated header");
    const uint32_t ver = (uint32_t)data[8] | ((uint32_t)data[9] << 8) | ((uint32_t)data[10] << 16) | ((uint32_t)data[11] << 24);
    if (ver != 211) return fail(ctx, PFV_ERR_VERSION, "codec version is not 2.1.1 (DecodeError::VersionError, src/dec.rs:57-59)");
    if (len < 20) return fail(ctx, PFV_ERR_IO, "truncated header");
    auto u16 = [&](size_t o) { return (int)data[o] | ((int)data[o + 1] << 8); };
    const int w = u16(12), h = u16(14), fps = u16(16), nq = u16(18);
    if (len < 20 + (size_t)nq * 128) return fail(ctx, PFV_ERR_IO, "truncated q-tables");
    std::vector<int32_t> q((size_t)std::max(nq, 1) * 64, 1);
    for (int i = 0; i < nq * 64; i++) q[(size_t)i] = u16(20 + 2 * (size_t)i);
    if (w > 0 && h > 0 && !(w & 1) && !(h & 1) && (uint64_t)max_gops * (uint64_t)pfv_total_blocks(w, h) * 256u > 0xffffffffull)
        return fail(ctx, PFV_ERR_BAD_ARG, "pfv_gop_decoder_create: max_gops x macroblocks x 256 exceeds the 32-bit coefficient index");
    pfv_dec_session *hot = nullptr;
    int rc = pfv_dec_session_create(ctx, w, h, q.data(), nq, max_gops, &hot);
    if (rc) return rc;
    pfv_gop_decoder *d = new pfv_gop_decoder();
    d->ctx = ctx; d->hot = hot; d->data = data; d->len = len;
    d->pos = d->reset_pos = 20 + (size_t)nq * 128;
    d->width = w; d->height = h; d->framerate = fps; d->n_qtables = nq; d->max_gops = max_gops; d->max_len = max_gop_frames;
    d->total_blocks = (size_t)pfv_total_blocks(w, h);
    d->frame_bytes = pfv_frame_bytes(w, h);
    d->cap = d->total_blocks * 256 / 4;                        // per slot: denser than 1 non-zero in 4 -> dense fallback
    const size_t S = (size_t)max_gops, tb = d->total_blocks;
    bool ok = true;
    hipError_t he = hipSuccess;
    for (GopDecSet &s : d->set) {
        ok = ok && s.idx.resize(S * d->cap) && s.val.resize(S * d->cap) && s.counts.resize(S) && s.mv.resize(S * tb * 2) && s.has.resize(S * tb) &&
             s.flags.resize(S);
        s.rc.assign(S, 0); s.qidx.assign(S * 3, 0); s.ev.assign(S, nullptr);
        if (ok) { memset(s.mv.data(), 0, S * tb * 2); memset(s.has.data(), 0, S * tb); memset(s.flags.data(), 0, S * sizeof(int)); }
        if (he == hipSuccess) he = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    }
    // the scatter kernel reads the coefficient lists where the parsers wrote them: that memory must be page-locked (= mapped to the device)
    bool lists_pinned = true;
    for (GopDecSet &s : d->set) lists_pinned = lists_pinned && s.idx.pinned && s.val.pinned && s.counts.pinned;
    ok = ok && d->frames_host.resize(S * (size_t)max_gop_frames * d->frame_bytes);
    if (ok && he == hipSuccess) he = hipMalloc((void **)&d->frames_dev, S * d->frame_bytes);
    if (!ok) rc = fail(ctx, PFV_ERR_NOMEM, "pfv_gop_decoder_create: host staging");
    else if (!lists_pinned) rc = fail(ctx, PFV_ERR_NOMEM, "pfv_gop_decoder_create: the coefficient-list staging could not be page-locked (locked-memory limit): use a smaller max_gops");
    else if (he != hipSuccess) rc = hip_fail(ctx, he, "pfv_gop_decoder_create");
    if (!rc) rc = dec_staging(hot);
    if (!rc) rc = pfv_dec_set_output_dev(hot, d->frames_dev);
    // the device-entropy path keeps the coefficient arrays of a whole batch in HBM (PFV_OPT_ENTROPY_DECODE)
    if (!rc && ctx->opt_entropy_decode != PFV_ENTROPY_DECODE_HOST && tb > 0) {
        GopDecDev &v = d->dev;
        const size_t F = S * (size_t)max_gop_frames;
        // per frame: motion vectors, flags, coded list, counts; the coefficient lists at 4 bytes per 3 payload bits at most (entd_pool_cap)
        const size_t list_guess = std::min(std::min(len, F * (tb * 512 / 8 + 64)) * 8 / 3 + F * 4, F * tb * 256);
        const size_t need = F * tb * (2 + 1 + 4 + 4 + 64) + list_guess * 4;
        size_t free_b = 0, total_b = 0;
        bool fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 2;
        if (ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE) fits = true;
        if (fits) {
            hipError_t e2 = hipSuccess;
            if (getenv("PFV_GOPD_WINDOW_STREAMS")) v.n_streams = std::max(1, std::min((int)GopDecDev::kStreams, atoi(getenv("PFV_GOPD_WINDOW_STREAMS"))));   // experiments
            for (int k = 0; k < v.n_streams && e2 == hipSuccess; k++) e2 = hipStreamCreateWithFlags(&v.streams[k], hipStreamNonBlocking);
            if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&v.up_stream, hipStreamNonBlocking);
            if (e2 == hipSuccess && v.lists.create(ctx, F, tb, list_guess) != PFV_OK) e2 = hipErrorOutOfMemory;
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&v.mv_dev, F * tb * 2);
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&v.has_dev, F * tb);
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&v.coded_dev, F * tb * 4);
            // a batch's payloads are at most the whole stream: size the staging now, not inside the first batch
            const size_t bytes_guess = std::min(len + F * 32 + 64, F * (tb * 512 / 8 + 64));
            v.sub_bits = (uint32_t)ctx->opt_entdec_lane_bits; v.launches = ctx->opt_entdec_launches; v.inner = ctx->opt_entdec_inner;   // PFV_OPT_ENTDEC_*
            const size_t sub_guess = bytes_guess * 8 / v.sub_bits + F;
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&v.bytes_dev, bytes_guess);
            if (e2 == hipSuccess) { v.bytes_cap = bytes_guess; e2 = hipMalloc((void **)&v.sub_dev, sub_guess * 4 * sizeof(uint32_t)); }
            if (e2 == hipSuccess) { v.sub_cap = sub_guess * 4; e2 = hipMalloc((void **)&v.groups_dev, (sub_guess / kEdOwn + F) * sizeof(uint2)); }
            if (e2 == hipSuccess) { v.groups_cap = sub_guess / kEdOwn + F; e2 = hipMalloc((void **)&v.wgsum_dev, v.groups_cap * sizeof(unsigned long long)); }
            if (e2 == hipSuccess) { v.wgsum_cap = v.groups_cap; e2 = hipMalloc((void **)&v.pk_dev, F * sizeof(EdPacket)); }
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&v.status_dev, F * sizeof(uint32_t));
            if (e2 == hipSuccess) v.pk_cap = F;
            const bool host_ok = e2 == hipSuccess && v.mv_host.resize(F * tb * 2) && v.has_host.resize(F * tb) && v.bytes_host.resize(bytes_guess) &&
                                 v.pk_host.resize(F) && v.status_host.resize(F) && v.groups_host.resize(sub_guess / kEdOwn + F) && v.flags_host.resize(F);
            if (host_ok) {
                memset(v.mv_host.data(), 0, F * tb * 2);
                memset(v.has_host.data(), 0, F * tb);
                v.frames_cap = F;
                v.on = true;
            } else {
                (void)hipGetLastError();
                for (void **p : {(void **)&v.hdr_maps_dev, (void **)&v.hdr_start_dev, (void **)&v.wgsum_dev, (void **)&v.mv_dev, (void **)&v.has_dev, (void **)&v.coded_dev, (void **)&v.bytes_dev, (void **)&v.sub_dev,
                                 (void **)&v.groups_dev, (void **)&v.pk_dev, (void **)&v.status_dev})
                    if (*p) { (void)hipFree(*p); *p = nullptr; }
                v.lists.destroy();
                v.bytes_cap = v.sub_cap = v.groups_cap = v.pk_cap = v.wgsum_cap = v.hdr_maps_cap = v.hdr_start_cap = 0;
                for (hipStream_t &st : v.streams)
                    if (st) { (void)hipStreamDestroy(st); st = nullptr; }
                if (v.up_stream) { (void)hipStreamDestroy(v.up_stream); v.up_stream = nullptr; }
                if (ctx->opt_entropy_decode == PFV_ENTROPY_DECODE_DEVICE)
                    rc = fail(ctx, PFV_ERR_NOMEM, "pfv_gop_decoder_create: the batch's coefficient arrays do not fit the device (PFV_ENTROPY_DECODE_DEVICE): use a smaller batch");
            }
        }
    }
    if (rc) { pfv_gop_decoder_destroy(d); return rc; }
    for (int t = 0; t < n_threads; t++) d->workers.emplace_back(gopd_worker, d);
    *out = d;
    return PFV_OK;
}
PFV_API int pfv_gop_decoder_width(const pfv_gop_decoder *d) { return d ? d->width : 0; }
PFV_API int pfv_gop_decoder_height(const pfv_gop_decoder *d) { return d ? d->height : 0; }
PFV_API int pfv_gop_decoder_framerate(const pfv_gop_decoder *d) { return d ? d->framerate : 0; }
PFV_API long pfv_gop_decoder_batches(const pfv_gop_decoder *d) { return d ? d->batches : 0; }
/* host seconds so far: out[0] header scan, [1] waiting for packet parsers, [2] waiting for the device before a staging set can be reused,
 * [3] enqueueing (incl. the time since the previous measurement point), [4] waiting for a batch's last frames, [5] waiting for the device's
 * entropy stage; counts: [6] packets whose payload the device read, [7] packets of device-entropy batches the host parser read, of which
 * [8] because the device's read had not settled and [9] because it found the payload irregular; [10] coefficient lists of host-parsed packets that outgrew
 * their place in the list pool and got a buffer of their own (one-symbol tables); returns entries written */
PFV_API int pfv_gop_decoder_stats(const pfv_gop_decoder *d, double *out, int n)
{
    if (!d || !out) return 0;
    const int k = std::min(n, 11);
    const double counts[5] = {(double)d->dev.packets_dev, (double)d->dev.packets_host, (double)d->dev.unsettled, (double)d->dev.irregular, (double)d->dev.lists.spilled};
    for (int i = 0; i < k; i++) out[i] = i < 6 ? d->stats[i] : counts[i - 6];
    return k;
}

// on != 0: decoded frames stay in device memory and the callback's y / u / v are DEVICE pointers (valid until the call that starts the next
// batch; ordered on the context's stream, which is idle when the callback runs) -- for consumers on the GPU (the reference README's
// texture-out wish); the download, the whole PCIe cost of decoding, is then not paid.  Only between batches (PFV_ERR_STATE otherwise).
PFV_API int pfv_gop_decoder_set_output_device(pfv_gop_decoder *d, int on)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    if (d->next_event < d->events.size()) return fail(d->ctx, PFV_ERR_STATE, "pfv_gop_decoder_set_output_device: a batch is being delivered");
    d->out_dev = on != 0;
    return PFV_OK;
}

// Decoder::reset (src/dec.rs:148-152).  The framebuffer is NOT rewound (neither is the reference's); unlike the frame-by-frame decoder
// this one has decoded ahead of the frames it delivered, so a stream whose first packet after the reset is a p-frame sees the state of
// the last DECODED frame, not of the last delivered one.  Streams start with an i-frame.
PFV_API int pfv_gop_decoder_reset(pfv_gop_decoder *d)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    d->eof = false;
    d->events.clear(); d->next_event = 0;
    d->pos = d->reset_pos;
    d->cont_valid = false;
    return PFV_OK;
}

// Decoder::advance_frame (src/dec.rs:169-224): 1 = Ok(true), 0 = Ok(false) (EOF), negative = error -- the same sequence of results, frames
// and callbacks as pfv_decoder_advance_frame on the same bytes.  y / u / v stay valid until the call that starts the next batch.
PFV_API int pfv_gop_decoder_advance_frame(pfv_gop_decoder *d, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    if (d->eof) return 0;
    if (d->next_event >= d->events.size()) {
        GopClock clk;
        gopd_scan_batch(d);
        d->stats[0] += clk.lap();
        int rc = d->dev.on ? gopd_decode_batch_dev(d) : 1;
        if (rc == 1) {
            if (d->dev.on) d->dev.batches_host++;
            rc = gopd_decode_batch(d);
        }
        if (rc) { gopd_drain_pool(d); d->events.clear(); d->next_event = 0; return rc; }
    }
    GopDecEvent &e = d->events[d->next_event];
    if (e.kind == GopDecEvent::END) { d->pos = e.pos_after; d->eof = true; return 0; }
    if (e.kind == GopDecEvent::ERROR) {   // the next call scans on from where the sequential loop would (src/dec.rs:174-182: the bytes read are gone)
        const int rc = e.rc;
        const char *msg = e.msg;
        d->pos = e.pos_after;
        d->events.clear(); d->next_event = 0;
        return fail(d->ctx, rc, msg);
    }
    d->next_event++;
    d->pos = e.pos_after;
    if (e.kind == GopDecEvent::DROP) return 1;
    if (e.rc) return fail(d->ctx, e.rc, e.msg);
    if (onvideo) {
        const uint8_t *f = (d->out_dev ? d->frames_all_dev : d->frames_host.data()) + ((size_t)e.t * (size_t)d->max_gops + (size_t)e.slot) * d->frame_bytes;
        const size_t ny = (size_t)d->width * d->height, nc = (size_t)(d->width / 2) * (d->height / 2);
        onvideo(user, f, f + ny, f + ny + nc, d->width, d->height);
    }
    return 1;
}

// Decoder::advance_delta (src/dec.rs:154-167)
PFV_API int pfv_gop_decoder_advance_delta(pfv_gop_decoder *d, double delta, pfv_video_cb onvideo, void *user)
{
    if (!d) return fail(nullptr, PFV_ERR_BAD_ARG, "null decoder");
    d->delta_accum += delta;
    const double delta_per_frame = 1.0 / (double)d->framerate;
    while (d->delta_accum >= delta_per_frame) {
        int rc = pfv_gop_decoder_advance_frame(d, onvideo, user);
        if (rc <= 0) return rc;
        d->delta_accum -= delta_per_frame;
    }
    return 1;
}

}  // extern "C"
