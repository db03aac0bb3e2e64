// Every THREADED object of the library from a native caller, for ThreadSanitizer (tools/sanitize.sh tsan; also run plain by
// tests/test_cpp_mirror.py): the reference gets data-race freedom from safe Rust (src/common.rs:411-413, 438-443: rayon over borrowed
// slices); here it is C++ worker threads behind a C ABI, so it is checked.
//   pfv_decoder         0 / 1 / 3 look-ahead threads x payloads read on the host / on the device (entropy windows on a second stream),
//                       on the intact stream and on damaged ones (byte flips, truncation): the parse threads run ahead of an error
//   pfv_gop_decoder     parse pool of 3 + device-entropy windows + the host-parser fallback of damaged packets, small batches (runs are cut)
//   pfv_gop_encoder     copy stream + batch collection
//   pfv_batch_encoder / pfv_batch_decoder   their pools (upload / collection thread; parsers)
// Check beside the sanitizer's: for every stream, ALL configurations of an object deliver the same frames (a hash per frame), the same
// number of them and the same final error code -- "per-call results are those of the sequential loop" (src/dec.rs:169-224).
// usage: threads_driver [n_damaged [only_stream]]     (no input files: the clip is generated here; only_stream: run that stream alone --
// under ThreadSanitizer one process per stream keeps the run inside the runtime's per-process limits on fibers and trace memory)
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "pfv_hip.hpp"

static uint64_t fnv(const std::vector<uint8_t> &v, uint64_t h)
{
    for (uint8_t b : v) { h ^= b; h *= 1099511628211ull; }
    return h;
}
struct Outcome {
    std::vector<uint64_t> frames;
    int error = 0;          // 0: clean end of stream; else the pfv::Error code that ended it
    bool operator==(const Outcome &o) const { return frames == o.frames && error == o.error; }
};
static uint32_t rnd(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

int main(int argc, char **argv)
{
    const int n_damaged = argc > 1 ? std::atoi(argv[1]) : 6, only_stream = argc > 2 ? std::atoi(argv[2]) : -1;
    const size_t w = 96, h = 64;
    const int n_frames = 12, gop = 4, fps = 30, quality = 5;
    try {
        pfv::Context ctx(0);
        // ---- the clip: a textured field that pans, with noise on part of it (coded and skipped macroblocks)
        std::vector<pfv::VideoFrame> clip;
        for (int t = 0; t < n_frames; t++) {
            pfv::VideoFrame f(w, h);
            uint32_t s = 12345u + 77u * (uint32_t)t;
            for (size_t y = 0; y < h; y++)
                for (size_t x = 0; x < w; x++) {
                    const size_t xs = x + 2 * t, ys = y + t;
                    int v = (int)((xs * 7 + ys * 5) & 127) + (int)(((xs >> 3) ^ (ys >> 3)) & 1) * 60;
                    if (((x >> 4) + (y >> 4) + t) % 3 == 0) v += (int)(rnd(s) % 33) - 16;
                    f.plane_y.pixels[y * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            for (size_t k = 0; k < f.plane_u.pixels.size(); k++) {
                f.plane_u.pixels[k] = (uint8_t)(128 + ((k + 3 * t) % 23));
                f.plane_v.pixels[k] = (uint8_t)(120 + ((k * 3 + t) % 17));
            }
            clip.push_back(f);
        }
        // ---- encoders: frame by frame, GOP-batched (copy stream + collection), batch (two streams)
        std::stringstream ss(std::ios::in | std::ios::out | std::ios::binary);
        {
            pfv::Encoder enc(ss, w, h, fps, quality, ctx);
            for (int t = 0; t < n_frames; t++) (t % gop == 0) ? enc.encode_iframe(clip[t]) : enc.encode_pframe(clip[t]);
        }
        const std::string bytes = ss.str();
        {
            std::stringstream gs(std::ios::in | std::ios::out | std::ios::binary);
            {
                pfv::GopEncoder ge(gs, w, h, fps, quality, ctx, 2, 3);
                for (int t = 0; t < n_frames; t++) (t % gop == 0) ? ge.encode_iframe(clip[t]) : ge.encode_pframe(clip[t]);
            }
            if (gs.str() != bytes) { std::fprintf(stderr, "GopEncoder bytes differ\n"); return 1; }
        }
        {
            std::stringstream w0(std::ios::in | std::ios::out | std::ios::binary), w1(std::ios::in | std::ios::out | std::ios::binary);
            {
                pfv::BatchEncoder be({&w0, &w1}, w, h, fps, quality, ctx);
                const size_t fb = be.frame_bytes(), ny = w * h, nc = (w / 2) * (h / 2);
                for (int t = 0; t < n_frames; t++) {
                    uint8_t *dst = be.frames();
                    for (int k = 0; k < 2; k++) {
                        std::copy(clip[t].plane_y.pixels.begin(), clip[t].plane_y.pixels.end(), dst + k * fb);
                        std::copy(clip[t].plane_u.pixels.begin(), clip[t].plane_u.pixels.end(), dst + k * fb + ny);
                        std::copy(clip[t].plane_v.pixels.begin(), clip[t].plane_v.pixels.end(), dst + k * fb + ny + nc);
                    }
                    (t % gop == 0) ? be.encode_iframes() : be.encode_pframes();
                }
            }
            if (w0.str() != bytes || w1.str() != bytes) { std::fprintf(stderr, "BatchEncoder bytes differ\n"); return 1; }
        }
        // ---- streams: the intact one, byte flips at seeded places behind the header, one truncation
        std::vector<std::string> streams{bytes};
        uint32_t seed = 99u;
        for (int k = 0; k < n_damaged; k++) {
            std::string d = bytes;
            if (k == n_damaged - 1) d.resize(d.size() * 2 / 3);
            else for (int j = 0; j < 1 + k % 3; j++) d[600 + rnd(seed) % (d.size() - 600)] ^= (char)(1u << (rnd(seed) % 8));
            streams.push_back(d);
        }
        auto hash_frame = [](const pfv::VideoFrame &fr) { return fnv(fr.plane_v.pixels, fnv(fr.plane_u.pixels, fnv(fr.plane_y.pixels, 1469598103934665603ull))); };
        int configs = 0;
        for (size_t si = 0; si < streams.size(); si++) {
            if (only_stream >= 0 && (int)si != only_stream) continue;
            std::vector<Outcome> outs;
            for (int mode : {PFV_ENTROPY_DECODE_HOST, PFV_ENTROPY_DECODE_DEVICE}) {
                ctx.check(pfv_ctx_set_option(ctx.handle(), PFV_OPT_ENTROPY_DECODE, mode));
                for (int la : {0, 1, 3}) {                                       // pfv_decoder: inline, 1 and 3 look-ahead threads
                    Outcome o;
                    try {
                        std::istringstream r(streams[si], std::ios::binary);
                        pfv::Decoder dec(r, ctx);
                        dec.set_lookahead(la);
                        while (dec.advance_frame([&](const pfv::VideoFrame &fr) { o.frames.push_back(hash_frame(fr)); })) {}
                    } catch (const pfv::Error &e) { o.error = e.code(); }
                    outs.push_back(o);
                    configs++;
                }
                for (int shape = 0; shape < 2; shape++) {                        // pfv_gop_decoder: parse pool of 3, batches of (2 x 3) and (8 x 15) frames
                    Outcome o;
                    try {
                        std::istringstream r(streams[si], std::ios::binary);
                        pfv::GopDecoder gd(r, ctx, shape ? 8 : 2, shape ? 15 : 3, 3);
                        while (gd.advance_frame([&](const pfv::VideoFrame &fr) { o.frames.push_back(hash_frame(fr)); })) {}
                    } catch (const pfv::Error &e) { o.error = e.code(); }
                    outs.push_back(o);
                    configs++;
                }
            }
            for (size_t k = 1; k < outs.size(); k++)
                if (!(outs[k] == outs[0])) {
                    std::fprintf(stderr, "stream %zu: configuration %zu delivered %zu frames / error %d, the sequential decoder %zu / %d\n", si, k, outs[k].frames.size(),
                                 outs[k].error, outs[0].frames.size(), outs[0].error);
                    return 1;
                }
            if (si == 0 && (outs[0].error != 0 || (int)outs[0].frames.size() != n_frames)) { std::fprintf(stderr, "intact stream: %zu frames, error %d\n", outs[0].frames.size(), outs[0].error); return 1; }
            std::printf("stream %zu (%s): %zu frames, final error %d -- identical under %zu configurations\n", si, si ? "damaged" : "intact", outs[0].frames.size(), outs[0].error, outs.size());
        }
        ctx.check(pfv_ctx_set_option(ctx.handle(), PFV_OPT_ENTROPY_DECODE, PFV_ENTROPY_DECODE_AUTO));
        // ---- batch decoder: pools of 2 and 3 parsers, two streams in lockstep
        for (int th : {2, 3})
            for (int mode : {PFV_ENTROPY_DECODE_HOST, PFV_ENTROPY_DECODE_DEVICE}) {
                ctx.check(pfv_ctx_set_option(ctx.handle(), PFV_OPT_ENTROPY_DECODE, mode));
                pfv::BatchDecoder bd({bytes, bytes}, ctx, th);
                const uint8_t *frames = nullptr;
                int steps = 0;
                while (bd.advance_frames(&frames) != 0) steps++;
                if (steps != n_frames) { std::fprintf(stderr, "BatchDecoder: %d steps\n", steps); return 1; }
                configs++;
            }
        std::printf("threads_driver OK: %zu streams, %d object configurations\n", streams.size(), configs);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
