// Exercise of include/pfv_hip.hpp (the C++ mirror of pfv_rs::enc::Encoder / dec::Decoder): reads raw 4:2:0 frames,
// writes the .pfv stream and the decoded frames.  usage: roundtrip W H FPS QUALITY GOP DROP_AT in.yuv out.pfv out.yuv
// (frame DROP_AT becomes a drop frame; -1: none).  The Python test compares both outputs with the oracle's.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <algorithm>
#include <vector>

#include "pfv_hip.hpp"

int main(int argc, char **argv)
{
    if (argc != 10) { std::fprintf(stderr, "usage: %s W H FPS QUALITY GOP DROP_AT in.yuv out.pfv out.yuv\n", argv[0]); return 2; }
    const size_t w = std::strtoul(argv[1], nullptr, 10), h = std::strtoul(argv[2], nullptr, 10);
    const unsigned fps = (unsigned)std::atoi(argv[3]);
    const int quality = std::atoi(argv[4]), gop = std::atoi(argv[5]), drop_at = std::atoi(argv[6]);
    try {
        pfv::Context ctx(0);
        if (const char *m = std::getenv("PFV_TEST_ENTROPY_DECODE"))      // tools/sanitize.sh: the decoders' device entropy stage on packets of any size
            ctx.check(pfv_ctx_set_option(ctx.handle(), PFV_OPT_ENTROPY_DECODE, std::atoi(m)));
        std::ifstream in(argv[7], std::ios::binary);
        std::stringstream stream(std::ios::in | std::ios::out | std::ios::binary);
        int n_in = 0;
        {
            pfv::Encoder enc(stream, w, h, fps, quality, ctx);
            pfv::VideoFrame f(w, h);
            for (int t = 0;; t++) {
                in.read(reinterpret_cast<char *>(f.plane_y.pixels.data()), (std::streamsize)f.plane_y.pixels.size());
                in.read(reinterpret_cast<char *>(f.plane_u.pixels.data()), (std::streamsize)f.plane_u.pixels.size());
                in.read(reinterpret_cast<char *>(f.plane_v.pixels.data()), (std::streamsize)f.plane_v.pixels.size());
                if (!in) break;
                if (t == drop_at) enc.encode_dropframe();
                else if (t % gop == 0) enc.encode_iframe(f);
                else enc.encode_pframe(f);
                n_in++;
            }
        }   // ~Encoder writes the EOF packet (impl Drop, src/enc.rs:28-34)
        const std::string bytes = stream.str();
        std::ofstream(argv[8], std::ios::binary).write(bytes.data(), (std::streamsize)bytes.size());

        std::istringstream reader(bytes, std::ios::binary);
        pfv::Decoder dec(reader, ctx);
        if (dec.width() != w || dec.height() != h || dec.framerate() != fps) { std::fprintf(stderr, "header mismatch\n"); return 1; }
        std::ofstream out(argv[9], std::ios::binary);
        int n_out = 0;
        auto sink = [&](const pfv::VideoFrame &fr) {
            out.write(reinterpret_cast<const char *>(fr.plane_y.pixels.data()), (std::streamsize)fr.plane_y.pixels.size());
            out.write(reinterpret_cast<const char *>(fr.plane_u.pixels.data()), (std::streamsize)fr.plane_u.pixels.size());
            out.write(reinterpret_cast<const char *>(fr.plane_v.pixels.data()), (std::streamsize)fr.plane_v.pixels.size());
            n_out++;
        };
        while (dec.advance_frame(sink)) {}
        // a bad stream must surface as pfv::Error with the DecodeError code
        try {
            std::istringstream bad(std::string("NOTPFV!!") + bytes.substr(8), std::ios::binary);
            pfv::Decoder d2(bad, ctx);
            std::fprintf(stderr, "bad magic accepted\n");
            return 1;
        } catch (const pfv::Error &e) {
            if (e.code() != PFV_ERR_FORMAT) { std::fprintf(stderr, "wrong error code %d\n", e.code()); return 1; }
        }
        // the GOP-batched objects on the same clip: same bytes, same frames (batches of 2 groups of at most 2 frames: runs are cut)
        {
            std::ifstream in3(argv[7], std::ios::binary);
            std::stringstream gs(std::ios::in | std::ios::out | std::ios::binary);
            {
                pfv::GopEncoder ge(gs, w, h, fps, quality, ctx, 2, 2);
                pfv::VideoFrame f(w, h);
                for (int t = 0; t < n_in; t++) {
                    in3.read(reinterpret_cast<char *>(f.plane_y.pixels.data()), (std::streamsize)f.plane_y.pixels.size());
                    in3.read(reinterpret_cast<char *>(f.plane_u.pixels.data()), (std::streamsize)f.plane_u.pixels.size());
                    in3.read(reinterpret_cast<char *>(f.plane_v.pixels.data()), (std::streamsize)f.plane_v.pixels.size());
                    if (t == drop_at) ge.encode_dropframe();
                    else if (t % gop == 0) ge.encode_iframe(f);
                    else ge.encode_pframe(f);
                }
            }   // ~GopEncoder flushes the open batch and writes the EOF packet
            if (gs.str() != bytes) { std::fprintf(stderr, "GopEncoder bytes differ from Encoder bytes\n"); return 1; }
            std::istringstream greader(bytes, std::ios::binary);
            pfv::GopDecoder gd(greader, ctx, 3, 2, 2);
            out.flush();
            std::ifstream ref(argv[9], std::ios::binary);
            const size_t ny = w * h, nc = (w / 2) * (h / 2);
            std::vector<char> want(ny + 2 * nc);
            int n_gop = 0;
            bool same = true;
            auto cmp = [&](const pfv::VideoFrame &fr) {
                ref.read(want.data(), (std::streamsize)want.size());
                same = same && ref && std::equal(fr.plane_y.pixels.begin(), fr.plane_y.pixels.end(), reinterpret_cast<const uint8_t *>(want.data())) &&
                       std::equal(fr.plane_u.pixels.begin(), fr.plane_u.pixels.end(), reinterpret_cast<const uint8_t *>(want.data()) + ny) &&
                       std::equal(fr.plane_v.pixels.begin(), fr.plane_v.pixels.end(), reinterpret_cast<const uint8_t *>(want.data()) + ny + nc);
                n_gop++;
            };
            while (gd.advance_frame(cmp)) {}
            if (!same || n_gop != n_out) { std::fprintf(stderr, "GopDecoder frames differ from Decoder's (%d of %d)\n", n_gop, n_out); return 1; }
            // the same stream with the frames left in device memory: fetched from the addresses the callback gets
            std::istringstream greader2(bytes, std::ios::binary);
            pfv::GopDecoder gd2(greader2, ctx, 3, 2, 2);
            gd2.set_output_device(true);
            ref.clear();
            ref.seekg(0);
            std::vector<uint8_t> got(ny + 2 * nc);
            int n_dev = 0;
            auto cmp_dev = [&](const uint8_t *y, const uint8_t *, const uint8_t *, uint32_t, uint32_t) {
                ref.read(want.data(), (std::streamsize)want.size());
                same = same && ref && pfv_dev_download(ctx.handle(), got.data(), y, got.size()) == PFV_OK &&
                       std::equal(got.begin(), got.end(), reinterpret_cast<const uint8_t *>(want.data()));
                n_dev++;
            };
            {   // the host-frame callbacks would read device pointers now: they must refuse, not crash (and leave the decoder where it was)
                bool refused = false;
                try { gd2.advance_frame([](const pfv::VideoFrame &) {}); } catch (const std::logic_error &) { refused = true; }
                try { gd2.advance_delta(1.0, [](const pfv::VideoFrame &) {}); refused = false; } catch (const std::logic_error &) {}
                if (!refused) { std::fprintf(stderr, "GopDecoder::advance_frame / advance_delta accepted a host callback with frames in device memory\n"); return 1; }
            }
            if (!gd2.advance_delta_device(1.0 / fps, cmp_dev)) { std::fprintf(stderr, "GopDecoder::advance_delta_device ended the stream early\n"); return 1; }
            while (gd2.advance_frame_device(cmp_dev)) {}
            if (!same || n_dev != n_out) { std::fprintf(stderr, "GopDecoder frames left in device memory differ (%d of %d)\n", n_dev, n_out); return 1; }
            // the encoder fed from DEVICE memory by reference (pfv_gop_encoder_set_frames_by_reference): the clip stays where it is, same bytes
            {
                const size_t fbytes = ny + 2 * nc, stride = (fbytes + 15) / 16 * 16;
                void *clip = nullptr;
                ctx.check(pfv_dev_alloc(ctx.handle(), stride * (size_t)n_in, &clip));
                std::ifstream in4(argv[7], std::ios::binary);
                std::vector<char> frame(fbytes);
                std::stringstream rs(std::ios::in | std::ios::out | std::ios::binary);
                {
                    pfv::GopEncoder ge(rs, w, h, fps, quality, ctx, 2, 2);
                    ge.set_frames_by_reference(true);
                    for (int t = 0; t < n_in; t++) {
                        in4.read(frame.data(), (std::streamsize)fbytes);
                        uint8_t *at = static_cast<uint8_t *>(clip) + stride * (size_t)t;
                        ctx.check(pfv_dev_upload(ctx.handle(), at, frame.data(), fbytes));
                        if (t == drop_at) ge.encode_dropframe();
                        else if (t % gop == 0) ge.encode_iframe_device(at);
                        else ge.encode_pframe_device(at);
                    }
                }   // ~GopEncoder: the last batch is collected here, the clip is still alive
                pfv_dev_free(ctx.handle(), clip);
                if (rs.str() != bytes) { std::fprintf(stderr, "GopEncoder reading device frames by reference wrote different bytes\n"); return 1; }
            }
            std::printf("gop: %d frames identical to the frame-by-frame objects\n", n_gop);
        }
        // the batch classes on the same clip: both streams carry the clip itself, so each writer must receive `bytes`
        // and every decoded step must equal the frames the single Decoder wrote (when no drop frame was asked for)
        if (drop_at < 0) {
            std::ifstream in2(argv[7], std::ios::binary);
            std::stringstream w0(std::ios::in | std::ios::out | std::ios::binary), w1(std::ios::in | std::ios::out | std::ios::binary);
            {
                pfv::BatchEncoder be({&w0, &w1}, w, h, fps, quality, ctx);
                const size_t fb = be.frame_bytes();
                std::vector<char> frame(fb);
                for (int t = 0; t < n_in; t++) {
                    in2.read(frame.data(), (std::streamsize)fb);
                    uint8_t *dst = be.frames();
                    std::copy(frame.begin(), frame.end(), reinterpret_cast<char *>(dst));
                    std::copy(frame.begin(), frame.end(), reinterpret_cast<char *>(dst) + fb);
                    if (t % gop == 0) be.encode_iframes();
                    else be.encode_pframes();
                }
            }   // ~BatchEncoder flushes and writes the EOF packets
            if (w0.str() != bytes || w1.str() != bytes) { std::fprintf(stderr, "BatchEncoder bytes differ from Encoder bytes\n"); return 1; }
            pfv::BatchDecoder bd({w0.str(), w1.str()}, ctx, 2);
            if (bd.width() != w || bd.height() != h) { std::fprintf(stderr, "batch header mismatch\n"); return 1; }
            out.flush();
            std::ifstream ref(argv[9], std::ios::binary);      // the frames the single Decoder just wrote
            const size_t fb = w * h * 3 / 2;
            std::vector<char> want(fb);
            const uint8_t *frames = nullptr;
            int steps = 0, rc;
            while ((rc = bd.advance_frames(&frames)) != 0) {
                if (rc != 1) { std::fprintf(stderr, "unexpected drop step\n"); return 1; }
                ref.read(want.data(), (std::streamsize)fb);
                if (!ref || !std::equal(want.begin(), want.end(), reinterpret_cast<const char *>(frames)) ||
                    !std::equal(want.begin(), want.end(), reinterpret_cast<const char *>(frames) + fb)) {
                    std::fprintf(stderr, "BatchDecoder frame %d differs from Decoder's\n", steps);
                    return 1;
                }
                steps++;
            }
            if (steps != n_out) { std::fprintf(stderr, "BatchDecoder decoded %d steps, Decoder %d\n", steps, n_out); return 1; }
            std::printf("batch: 2 streams x %d steps identical to the single-stream objects\n", steps);
        }
        std::printf("frames in %d, decoded %d, stream %zu bytes\n", n_in, n_out, bytes.size());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
