// e2e_native.cpp -- BASELINE config 4 end to end from a NATIVE host program (the reference's callers are Rust; the Python mirror pays
// ~0.15 ms of interpreter per delivered frame, more than the device needs for the frame): synthetic frames in page-locked memory ->
// pfv_gop_encoder -> .pfv bytes -> pfv_gop_decoder -> frames, through the C ABI only (include/pfv_hip.h).
//   g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$PWD/pretty-fast-video_amd -o /tmp/e2e_native
//   /tmp/e2e_native [width height frames gop quality enc_gops dec_gops parse_threads [lane_bits]]
// Prints one JSON object.  Measurement tool (bench.py runs it for extra.config4.end_to_end.native_host); not part of the library.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pfv_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(expr)                                                                                                   \
    do {                                                                                                              \
        int rc__ = (expr);                                                                                            \
        if (rc__ < 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc__, pfv_last_error(ctx)); return 1; }             \
    } while (0)

struct Sink {
    pfv_ctx *ctx;
    int w, h, every;
    bool device;
    long n = 0;
    uint64_t hash = 1469598103934665603ull, touched = 0;
    std::vector<uint8_t> tmp;
    uint8_t *kept = nullptr;     // frames left in HBM: the sampled frames are copied device-to-device (what a consumer on the GPU would do with
    int n_kept = 0, cap_kept = 0; // a frame it wants to keep) and fetched for the checksum when the clock has stopped
};
static void fnv(uint64_t &h, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i + 8 <= n; i += 8 * 61) {   // every 61st 8-byte word: a checksum of the sampled frames, not a benchmark of hashing
        uint64_t v;
        memcpy(&v, p + i, 8);
        h = (h ^ v) * 1099511628211ull;
    }
}
static void on_video(void *user, const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h)
{
    Sink *s = (Sink *)user;
    const size_t ny = (size_t)w * h, nc = (size_t)(w / 2) * (h / 2);
    if (s->n % s->every == 0) {
        if (s->device) {
            if (s->n_kept < s->cap_kept) pfv_dev_copy(s->ctx, s->kept + (size_t)s->n_kept++ * (ny + 2 * nc), y, ny + 2 * nc);    // the three planes are contiguous
        } else {
            fnv(s->hash, y, ny & ~(size_t)7);
            fnv(s->hash, u, nc & ~(size_t)7);
            fnv(s->hash, v, nc & ~(size_t)7);
        }
    } else if (!s->device) {
        s->touched += y[0] + u[0] + v[0] + y[ny - 1];                   // the consumer looks at the frame it is handed
    }
    s->n++;
}

// the frames a device-mode sink kept: down, into the checksum (outside the timed region), buffer freed
static void finish_kept(Sink &s, size_t fb, size_t ny, size_t nc)
{
    if (!s.kept) return;
    s.tmp.resize(fb);
    for (int k = 0; k < s.n_kept; k++) {
        pfv_dev_download(s.ctx, s.tmp.data(), s.kept + (size_t)k * fb, fb);
        fnv(s.hash, s.tmp.data(), ny & ~(size_t)7);
        fnv(s.hash, s.tmp.data() + ny, nc & ~(size_t)7);
        fnv(s.hash, s.tmp.data() + ny + nc, nc & ~(size_t)7);
    }
    pfv_dev_free(s.ctx, s.kept);
    s.kept = nullptr;
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160, N = argc > 3 ? atoi(argv[3]) : 300;
    const int GOP = argc > 4 ? atoi(argv[4]) : 15, Q = argc > 5 ? atoi(argv[5]) : 5, EG = argc > 6 ? atoi(argv[6]) : 10, DG = argc > 7 ? atoi(argv[7]) : 20;
    const int threads = argc > 8 ? atoi(argv[8]) : 15, lane_bits = argc > 9 ? atoi(argv[9]) : 0;
    pfv_ctx *ctx = nullptr;
    if (pfv_ctx_create(0, &ctx) != PFV_OK) { fprintf(stderr, "no device: %s\n", pfv_last_error(nullptr)); return 1; }
    if (lane_bits) CHECK(pfv_ctx_set_option(ctx, PFV_OPT_ENTDEC_LANE_BITS, lane_bits));
    if (getenv("PFV_E2E_INNER")) CHECK(pfv_ctx_set_option(ctx, PFV_OPT_ENTDEC_INNER_ROUNDS, atoi(getenv("PFV_E2E_INNER"))));       // experiments: settling rounds of k_entd_sync
    if (getenv("PFV_E2E_LAUNCHES")) CHECK(pfv_ctx_set_option(ctx, PFV_OPT_ENTDEC_LAUNCHES, atoi(getenv("PFV_E2E_LAUNCHES"))));
    // The GOP encoders of this program hold 6-22 GB of device memory each.  They are destroyed when the encode section is over, not between its
    // timed passes: after the hipFree of a multi-GB buffer, device-to-host copies on most streams run at half the link's rate for the life of
    // the next object (24-29 instead of 55 GB/s; bisected inside pfv_gop_encoder_destroy with a probe copy), and a pass timed
    // behind it measures that, not the object (measured both ways on one box: 173-176 M / 1.04-1.08 G macroblocks/s this way, 132 M / 0.90-0.91 G
    // with every object destroyed before the next is created).  An application's encoder is long-lived.  PFV_E2E_DESTROY_EAGERLY=1: the old order.
    // (The decoders ARE destroyed one by one: keeping a dozen idle objects' streams alive makes the runtime share hardware queues between the
    // streams of the next one -- its entropy windows then wait behind the frame downloads, 51 ms instead of 7.)
    const bool eager = getenv("PFV_E2E_DESTROY_EAGERLY") != nullptr;
    std::vector<pfv_gop_encoder *> old_encoders;
    auto retire = [&](pfv_gop_encoder *e) { if (eager) pfv_gop_encoder_destroy(e); else old_encoders.push_back(e); };
    const size_t fb = pfv_frame_bytes(W, H), ny = (size_t)W * H, nc = (size_t)(W / 2) * (H / 2);
    const long n_mb = pfv_total_blocks(W, H);
    // the producer's frames, page-locked
    uint8_t *frames = nullptr;
    void *dev = nullptr;
    CHECK(pfv_host_alloc(ctx, fb * (size_t)N, (void **)&frames));
    CHECK(pfv_dev_alloc(ctx, fb, &dev));
    const uint64_t seed = 0x50465632ull;
    for (int t = 0; t < N; t++) {
        CHECK(pfv_synth_frames_dev(ctx, W, H, 1, &seed, t, (uint8_t *)dev));
        CHECK(pfv_dev_download(ctx, frames + (size_t)t * fb, dev, fb));
    }
    // ---- pfv_gop_encoder with the frames already in device memory (a renderer's output): nothing crosses PCIe on the way in.  Timed first as
    // the FIRST object of the process (behind a 64x48 warm-up object), then by batch width further down.
    uint8_t *all_dev = nullptr;
    CHECK(pfv_dev_alloc(ctx, fb * (size_t)N, (void **)&all_dev));
    for (int t = 0; t < N; t++) CHECK(pfv_synth_frames_dev(ctx, W, H, 1, &seed, t, all_dev + (size_t)t * fb));
    CHECK(pfv_ctx_sync(ctx));
    bool hbm_by_ref = false;          // pfv_gop_encoder_set_frames_by_reference: the batch's kernels read the clip where it lies, no copy
    auto hbm_pass = [&](int G, double *seconds, double *stats, size_t *bytes) -> int {
        pfv_gop_encoder *e = nullptr;
        CHECK(pfv_gop_encoder_create(ctx, W, H, 30, Q, G, GOP, 0, &e));
        if (hbm_by_ref) CHECK(pfv_gop_encoder_set_frames_by_reference(e, 1));
        size_t total = 0;
        auto drain = [&]() -> int {
            const pfv_iovec *iov = nullptr;
            size_t cnt = 0;
            int rc = pfv_gop_encoder_drain_iov(e, &iov, &cnt);
            for (size_t i = 0; !rc && i < cnt; i++) total += iov[i].len;
            return rc;
        };
        const double t0 = now();
        CHECK(drain());
        for (int t = 0; t < N; t++) {
            const uint8_t *f = all_dev + (size_t)t * fb;
            CHECK(t % GOP == 0 ? pfv_gop_encoder_encode_iframe_dev(e, f) : pfv_gop_encoder_encode_pframe_dev(e, f));
            CHECK(drain());
        }
        CHECK(pfv_gop_encoder_finish(e));
        CHECK(drain());
        *seconds = now() - t0;
        pfv_gop_encoder_stats(e, stats, 7);
        retire(e);
        *bytes = total;
        return 0;
    };
    double t_enc_hbm_first = 0, enc_hbm_first_stats[7] = {0, 0, 0, 0, 0, 0, 0};
    size_t hbm_first_total = 0;
    {
        {   // a warm-up object at 64x48 first (code objects, first launches): a few hundred KB, it leaves no such trace
            pfv_gop_encoder *e = nullptr;
            void *tiny = nullptr;
            CHECK(pfv_dev_alloc(ctx, pfv_frame_bytes(64, 48), &tiny));
            CHECK(pfv_gop_encoder_create(ctx, 64, 48, 30, Q, 2, GOP, 0, &e));
            for (int t = 0; t < 2 * GOP; t++) {
                CHECK(pfv_synth_frames_dev(ctx, 64, 48, 1, &seed, t, (uint8_t *)tiny));
                CHECK(t % GOP == 0 ? pfv_gop_encoder_encode_iframe_dev(e, (const uint8_t *)tiny) : pfv_gop_encoder_encode_pframe_dev(e, (const uint8_t *)tiny));
            }
            CHECK(pfv_gop_encoder_finish(e));
            pfv_gop_encoder_destroy(e);
            pfv_dev_free(ctx, tiny);
        }
        const int rc = hbm_pass(getenv("PFV_E2E_HBM_GOPS") ? atoi(getenv("PFV_E2E_HBM_GOPS")) : (EG > DG ? EG : DG), &t_enc_hbm_first, enc_hbm_first_stats, &hbm_first_total);
        if (rc) return rc;
    }
    // ---- encode: a timed pass whose writer only counts the segments, then one that keeps the bytes
    std::vector<uint8_t> stream;
    double t_enc = 0, enc_stats[5] = {0, 0, 0, 0, 0};
    for (int pass = 0; pass < 2; pass++) {
        pfv_gop_encoder *e = nullptr;
        CHECK(pfv_gop_encoder_create(ctx, W, H, 30, Q, EG, GOP, 0, &e));
        size_t total = 0;
        auto drain = [&]() -> int {
            const pfv_iovec *iov = nullptr;
            size_t cnt = 0;
            int rc = pfv_gop_encoder_drain_iov(e, &iov, &cnt);
            if (rc) return rc;
            for (size_t i = 0; i < cnt; i++) {
                total += iov[i].len;
                if (pass == 1) stream.insert(stream.end(), (const uint8_t *)iov[i].data, (const uint8_t *)iov[i].data + iov[i].len);
            }
            return 0;
        };
        const double t0 = now();
        CHECK(drain());
        for (int t = 0; t < N; t++) {
            const uint8_t *f = frames + (size_t)t * fb;
            CHECK(t % GOP == 0 ? pfv_gop_encoder_encode_iframe(e, f, f + ny, f + ny + nc) : pfv_gop_encoder_encode_pframe(e, f, f + ny, f + ny + nc));
            CHECK(drain());
        }
        CHECK(pfv_gop_encoder_finish(e));
        CHECK(drain());
        if (pass == 0) { t_enc = now() - t0; pfv_gop_encoder_stats(e, enc_stats, 5); }
        retire(e);
        if (pass == 1 && total != stream.size()) return 2;
    }
    // ---- the widths of the frames-in-HBM encoder (objects created after others were destroyed: see hbm_pass above)
    double t_enc_hbm = t_enc_hbm_first, enc_hbm_stats[5];
    for (int i = 0; i < 5; i++) enc_hbm_stats[i] = enc_hbm_first_stats[i];
    int hbm_gops = EG > DG ? EG : DG;
    std::string hbm_by_width, hbm_ref_by_width;
    double t_enc_hbm_ref = 0, ref_stats[7] = {0, 0, 0, 0, 0, 0, 0};
    {
        // the encoder copies a device frame on the caller's stream and runs its kernels on a stream of its own: with the clip in ONE batch the
        // copies (2.5 ms for 300 4K frames) stand in front of the kernels, with two or more they run under the kernels of the batch before;
        // narrower launches cost kernel efficiency.  Every width is timed (two passes each, the better one counts).
        const int whole = EG > DG ? EG : DG;
        int widths[3] = {whole, (whole + 1) / 2, (whole + 3) / 4};
        if (getenv("PFV_E2E_HBM_GOPS")) { widths[0] = atoi(getenv("PFV_E2E_HBM_GOPS")); widths[1] = widths[2] = 0; }
        for (int wi = 0; wi < 3; wi++) {
            const int G = widths[wi];
            if (G <= 0 || (wi && G == widths[wi - 1])) continue;
            double best = 0, best_stats[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int pass = 0; pass < 2; pass++) {
                double dt = 0, st[7];
                size_t total = 0;
                const int rc = hbm_pass(G, &dt, st, &total);
                if (rc) return rc;
                if (pass == 0 || dt < best) { best = dt; for (int i = 0; i < 5; i++) best_stats[i] = st[i]; }
                if (total != stream.size()) { fprintf(stderr, "encoder fed from device memory wrote %zu bytes, from host memory %zu\n", total, stream.size()); return 7; }
            }
            char b[96];
            snprintf(b, sizeof b, "%s\"%d\": %.1f", hbm_by_width.empty() ? "" : ", ", G, (double)N * n_mb / best);
            hbm_by_width += b;
            if (best < t_enc_hbm) { t_enc_hbm = best; hbm_gops = G; for (int i = 0; i < 5; i++) enc_hbm_stats[i] = best_stats[i]; }
        }
        if (hbm_first_total != stream.size()) { fprintf(stderr, "encoder fed from device memory wrote %zu bytes, from host memory %zu\n", hbm_first_total, stream.size()); return 7; }
        // the same widths with the frames taken BY REFERENCE (the clip stays where it is: no device-to-device copy at all)
        hbm_by_ref = true;
        for (int wi = 0; wi < 2; wi++) {      // the whole clip per batch (two passes) and half of it (one): every object of this section stays alive until its end (above)
            const int G = widths[wi];
            if (G <= 0 || (wi && G == widths[wi - 1])) continue;
            double best = 0;
            for (int pass = 0; pass < (wi == 0 ? 2 : 1); pass++) {
                double dt = 0, st[7];
                size_t total = 0;
                const int rc = hbm_pass(G, &dt, st, &total);
                if (rc) return rc;
                if (pass == 0 || dt < best) best = dt;
                if (t_enc_hbm_ref == 0 || dt < t_enc_hbm_ref) for (int i = 0; i < 7; i++) ref_stats[i] = st[i];
                if (t_enc_hbm_ref == 0 || dt < t_enc_hbm_ref) t_enc_hbm_ref = dt;
                if (total != stream.size()) { fprintf(stderr, "encoder reading device frames by reference wrote %zu bytes, from host memory %zu\n", total, stream.size()); return 7; }
            }
            char b[96];
            snprintf(b, sizeof b, "%s\"%d\": %.1f", hbm_ref_by_width.empty() ? "" : ", ", G, (double)N * n_mb / best);
            hbm_ref_by_width += b;
            if (t_enc_hbm_ref == 0 || best < t_enc_hbm_ref) t_enc_hbm_ref = best;
        }
        hbm_by_ref = false;
        pfv_dev_free(ctx, all_dev);
    }
    for (pfv_gop_encoder *e : old_encoders) pfv_gop_encoder_destroy(e);
    old_encoders.clear();
    if (getenv("PFV_E2E_STOP_AFTER_ENCODE")) { printf("{\"encode_value\": %.1f, \"encode_value_frames_in_hbm\": %.1f, \"first_object\": %.1f, \"by_width\": {%s}, \"quality\": %d, \"stream_bytes\": %zu, \"by_reference\": %.1f, \"by_reference_s\": %.5f, \"by_reference_by_width\": {%s}, \"by_reference_host_seconds\": {\"enqueue_s\": %.5f, \"kernel_wait_s\": %.5f, \"payload_download_s\": %.5f, \"packet_assembly_s\": %.5f, \"frames_by_reference\": %.0f, \"batches_redone\": %.0f}}\n", (double)N * n_mb / t_enc, (double)N * n_mb / t_enc_hbm, (double)N * n_mb / t_enc_hbm_first, hbm_by_width.c_str(), Q, stream.size(), (double)N * n_mb / t_enc_hbm_ref, t_enc_hbm_ref, hbm_ref_by_width.c_str(), ref_stats[1], ref_stats[2], ref_stats[3], ref_stats[4], ref_stats[5], ref_stats[6]); return 0; }   // timelines, quality sweeps
    // ---- decode
    struct Mode { const char *name; int entropy; bool device_out; };
    const Mode modes[] = {{"payloads_read_on_host", PFV_ENTROPY_DECODE_HOST, false},
                          {"payloads_read_on_device", PFV_ENTROPY_DECODE_DEVICE, false},
                          {"payloads_read_on_device_frames_left_in_hbm", PFV_ENTROPY_DECODE_DEVICE, true}};
    std::string out = "{";
    char buf[8192];
    snprintf(buf, sizeof buf,
             "\"workload\": \"%dx%d, %d frames, GOP-%d, quality %d\", \"stream_bytes\": %zu, \"encode_value\": %.1f, \"encode_value_frames_in_hbm\": %.1f, \"encode_s\": %.5f, "
             "\"encoder_host_seconds\": {\"upload_wait_s\": %.5f, \"enqueue_s\": %.5f, \"kernel_wait_s\": %.5f, \"payload_download_s\": %.5f, \"packet_assembly_s\": %.5f}, "
             "\"encode_frames_in_hbm_s\": %.5f, \"encoder_host_seconds_frames_in_hbm\": {\"upload_wait_s\": %.5f, \"enqueue_s\": %.5f, \"kernel_wait_s\": %.5f, \"payload_download_s\": %.5f, \"packet_assembly_s\": %.5f}, "
             "\"encode_value_frames_in_hbm_first_object_of_the_process\": %.1f, \"encode_value_frames_in_hbm_by_gops_per_batch\": {%s}, \"encode_value_frames_in_hbm_by_reference\": %.1f, \"encode_value_frames_in_hbm_by_reference_by_gops_per_batch\": {%s}, \"gops_per_batch\": {\"encoder\": %d, \"encoder_frames_in_hbm\": %d, \"decoder\": %d}, \"parse_threads\": %d, \"decode\": {",
             W, H, N, GOP, Q, stream.size(), (double)N * n_mb / t_enc, (double)N * n_mb / t_enc_hbm, t_enc, enc_stats[0], enc_stats[1], enc_stats[2], enc_stats[3], enc_stats[4],
             t_enc_hbm, enc_hbm_stats[0], enc_hbm_stats[1], enc_hbm_stats[2], enc_hbm_stats[3], enc_hbm_stats[4], (double)N * n_mb / t_enc_hbm_first, hbm_by_width.c_str(), t_enc_hbm_ref > 0 ? (double)N * n_mb / t_enc_hbm_ref : 0.0, hbm_ref_by_width.c_str(), EG, hbm_gops, DG, threads);
    out += buf;
    uint64_t want_hash = 0;
    const char *only = getenv("PFV_E2E_ONLY");        // profiling runs: one decode mode by name, nothing behind it
    for (size_t m = 0; m < sizeof modes / sizeof modes[0]; m++) {
        if (only && strcmp(only, modes[m].name) != 0) continue;
        double best = 1e30, st[10] = {0};
        for (int rep = 0; rep < 3; rep++) {   // the first run of a mode pays for code objects and first-touch of its buffers
            CHECK(pfv_ctx_set_option(ctx, PFV_OPT_ENTROPY_DECODE, modes[m].entropy));
            pfv_gop_decoder *d = nullptr;
            CHECK(pfv_gop_decoder_create(ctx, stream.data(), stream.size(), DG, GOP, threads, &d));
            CHECK(pfv_gop_decoder_set_output_device(d, modes[m].device_out ? 1 : 0));
            Sink s{ctx, W, H, 101, modes[m].device_out};
            if (modes[m].device_out) { s.cap_kept = N / 101 + 1; CHECK(pfv_dev_alloc(ctx, fb * (size_t)s.cap_kept, (void **)&s.kept)); }
            const double t0 = now();
            int rc;
            while ((rc = pfv_gop_decoder_advance_frame(d, on_video, &s)) == 1) {}
            if (modes[m].device_out) CHECK(pfv_ctx_sync(ctx));           // the kept frames' copies are part of the consumer's work
            const double el = now() - t0;
            CHECK(rc);
            finish_kept(s, fb, ny, nc);
            if (s.n != N) { fprintf(stderr, "%s: %ld frames of %d\n", modes[m].name, s.n, N); return 3; }
            if (!want_hash) want_hash = s.hash;
            if (s.hash != want_hash) { fprintf(stderr, "%s: decoded frames differ between modes\n", modes[m].name); return 4; }
            if (el < best) { best = el; pfv_gop_decoder_stats(d, st, 10); }
            pfv_gop_decoder_destroy(d);
        }
        snprintf(buf, sizeof buf,
                 "%s\"%s\": {\"decode_value\": %.1f, \"decode_s\": %.5f, \"decoder_host_seconds\": {\"scan_s\": %.5f, \"parse_wait_s\": %.5f, \"device_wait_s\": %.5f, "
                 "\"enqueue_s\": %.5f, \"final_wait_s\": %.5f, \"device_entropy_wait_s\": %.5f}, \"packets_read_on_device\": %.0f, \"packets_left_to_host_parser\": %.0f, \"left_unsettled\": %.0f, \"left_irregular\": %.0f}",
                 (m && !only) ? ", " : "", modes[m].name, (double)N * n_mb / best, best, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7], st[8], st[9]);
        out += buf;
    }
    if (only) { out += "}}"; puts(out.c_str()); return 0; }
    // ---- the frame-by-frame objects (the reference's own call pattern: one packet per call, src/enc.rs:75-173, src/dec.rs:169-224)
    double t_senc = 0, t_sdec = 0, t_sdec_host = 0, t_sdec_hbm = 0, t_sdec0 = 0, t_sdec_hbm0 = 0;
    long sdec_counts[2] = {0, 0};
    {
        pfv_encoder *e = nullptr;
        CHECK(pfv_encoder_create(ctx, W, H, 30, Q, &e));
        size_t total = 0;
        const double t0 = now();
        for (int t = 0; t < N; t++) {
            const uint8_t *f = frames + (size_t)t * fb;
            CHECK(t % GOP == 0 ? pfv_encoder_encode_iframe(e, f, f + ny, f + ny + nc) : pfv_encoder_encode_pframe(e, f, f + ny, f + ny + nc));
            const uint8_t *p = nullptr;
            size_t n = 0;
            CHECK(pfv_encoder_drain(e, &p, &n));
            total += n;
        }
        CHECK(pfv_encoder_finish(e));
        const uint8_t *p = nullptr;
        size_t n = 0;
        CHECK(pfv_encoder_drain(e, &p, &n));
        total += n;
        t_senc = now() - t0;
        pfv_encoder_destroy(e);
        if (total != stream.size()) { fprintf(stderr, "frame-by-frame encoder wrote %zu bytes, the GOP-batched one %zu\n", total, stream.size()); return 5; }
        for (int mode = 0; mode < 5; mode++) {   // run streams read by the host parser (look-ahead threads) / by the device stage / + frames left in HBM; 3, 4 = 1, 2 without parser threads
            const bool no_threads = mode >= 3;
            const int m3 = no_threads ? mode - 2 : mode;
            CHECK(pfv_ctx_set_option(ctx, PFV_OPT_ENTROPY_DECODE, m3 ? PFV_ENTROPY_DECODE_AUTO : PFV_ENTROPY_DECODE_HOST));
            double best = 1e30;
            for (int rep = 0; rep < 2; rep++) {
                pfv_decoder *d = nullptr;
                CHECK(pfv_decoder_create(ctx, stream.data(), stream.size(), &d));
                if (getenv("PFV_E2E_LOOKAHEAD")) CHECK(pfv_decoder_set_lookahead(d, atoi(getenv("PFV_E2E_LOOKAHEAD"))));     // experiments
                if (no_threads) CHECK(pfv_decoder_set_lookahead(d, 0));
                CHECK(pfv_decoder_set_output_device(d, m3 == 2 ? 1 : 0));
                Sink s{ctx, W, H, 101, m3 == 2};
                if (m3 == 2) { s.cap_kept = N / 101 + 1; CHECK(pfv_dev_alloc(ctx, fb * (size_t)s.cap_kept, (void **)&s.kept)); }
                const double t1 = now();
                int rc;
                while ((rc = pfv_decoder_advance_frame(d, on_video, &s)) == 1) {}
                if (m3 == 2) CHECK(pfv_ctx_sync(ctx));
                const double el = now() - t1;
                CHECK(rc);
                finish_kept(s, fb, ny, nc);
                pfv_decoder_entropy_counts(d, sdec_counts);
                pfv_decoder_destroy(d);
                if (s.n != N || s.hash != want_hash) { fprintf(stderr, "frame-by-frame decoder: %ld frames, hash %s\n", s.n, s.hash == want_hash ? "ok" : "differs"); return 6; }
                best = el < best ? el : best;
            }
            (mode == 4 ? t_sdec_hbm0 : mode == 3 ? t_sdec0 : mode == 2 ? t_sdec_hbm : mode ? t_sdec : t_sdec_host) = best;
        }
    }
    snprintf(buf, sizeof buf, "}, \"frame_by_frame_objects\": {\"encode_value\": %.1f, \"decode_value\": %.1f, \"decode_value_frames_left_in_hbm\": %.1f, \"decode_value_payloads_read_on_host\": %.1f, "
             "\"decode_value_no_parser_threads\": %.1f, \"decode_value_no_parser_threads_frames_left_in_hbm\": %.1f, \"packets_read_on_device\": %ld, \"packets_left_to_host_parser\": %ld, \"note\": \"pfv_encoder / pfv_decoder, one packet per call "
             "(default look-ahead of 4 threads; no_parser_threads: pfv_decoder_set_lookahead(d, 0) -- the block headers are read on the device, the caller's thread stages the next payload); the same bytes and frames\"", (double)N * n_mb / t_senc, (double)N * n_mb / t_sdec, (double)N * n_mb / t_sdec_hbm, (double)N * n_mb / t_sdec_host,
             (double)N * n_mb / t_sdec0, (double)N * n_mb / t_sdec_hbm0, sdec_counts[0], sdec_counts[1]);
    out += buf;
    out += "}";
    out += ", \"frames_checked\": \"every 101st frame sampled (every 61st word) in every mode: identical\"}";
    puts(out.c_str());
    pfv_host_free(ctx, frames);
    pfv_ctx_destroy(ctx);
    return 0;
}
