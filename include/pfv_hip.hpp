// pfv_hip.hpp -- C++ host classes over the C ABI of pfv_hip.h, with the reference's public surface:
//   pfv::VideoPlane  (src/plane.rs:1-36)     width, height, pixels  (public fields, tightly packed rows)
//   pfv::VideoFrame  (src/frame.rs:3-59)     width, height, plane_y / plane_u / plane_v
//   pfv::Encoder     (src/enc.rs:12-188)     new(writer, w, h, framerate, quality, num_threads) -> Encoder(writer, ..., Context&)
//   pfv::Decoder     (src/dec.rs:15-224)     new(reader, num_threads)                            -> Decoder(reader, Context&)
//   pfv::GopEncoder / pfv::GopDecoder        the same two objects with the independent GOPs of the stream as the slots of every launch
//                                            (pfv_gop_encoder / pfv_gop_decoder): same bytes, same frames, same results call by call
// The reference's `num_threads` slot (its rayon pool) is the pfv::Context: one device + one HIP stream.  Writers are
// std::ostream, readers std::istream (read to the end on construction; the reference needs Read + Seek).  Errors of the
// C ABI become pfv::Error (what() = pfv_last_error); DecodeError::{FormatError, VersionError, IOError} keep their codes.
// Header-only; link against libpfv_hip.so.
#pragma once

#include <cstdint>
#include <functional>
#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pfv_hip.h"

namespace pfv {

class Error : public std::runtime_error {
  public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }   // a pfv_status value

  private:
    int code_;
};

class Context {
  public:
    explicit Context(int device = 0)
    {
        int rc = pfv_ctx_create(device, &h_);
        if (rc != PFV_OK) throw Error(rc, last(nullptr));
    }
    ~Context() { pfv_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    pfv_ctx *handle() const { return h_; }
    void check(int rc) const
    {
        if (rc != PFV_OK) throw Error(rc, last(h_));
    }

  private:
    static std::string last(pfv_ctx *h)
    {
        const char *m = pfv_last_error(h);
        return m ? m : "pfv_hip error";
    }
    pfv_ctx *h_ = nullptr;
};

// src/plane.rs:1-36
struct VideoPlane {
    size_t width = 0, height = 0;
    std::vector<uint8_t> pixels;
    VideoPlane() = default;
    VideoPlane(size_t w, size_t h) : width(w), height(h), pixels(w * h, 0) {}                    // VideoPlane::new
    static VideoPlane from_slice(size_t w, size_t h, const uint8_t *data, size_t len)           // VideoPlane::from_slice
    {
        if (len != w * h) throw std::invalid_argument("VideoPlane::from_slice: len != width * height (src/plane.rs:14)");
        VideoPlane p;
        p.width = w; p.height = h;
        p.pixels.assign(data, data + len);
        return p;
    }
};

// src/frame.rs:3-26
struct VideoFrame {
    size_t width = 0, height = 0;
    VideoPlane plane_y, plane_u, plane_v;
    VideoFrame() = default;
    VideoFrame(size_t w, size_t h) : width(w), height(h), plane_y(w, h), plane_u(w / 2, h / 2), plane_v(w / 2, h / 2)   // VideoFrame::new
    {
        if (w % 2 || h % 2) throw std::invalid_argument("VideoFrame: width and height must be even (src/frame.rs:13)");
        plane_u.pixels.assign(plane_u.pixels.size(), 128);
        plane_v.pixels.assign(plane_v.pixels.size(), 128);
    }
};

// src/enc.rs:12-188
class Encoder {
  public:
    Encoder(std::ostream &writer, size_t width, size_t height, uint32_t framerate, int quality, Context &ctx)
        : ctx_(ctx), out_(writer), width_(width), height_(height)
    {
        ctx_.check(pfv_encoder_create(ctx.handle(), (int)width, (int)height, (int)framerate, quality, &h_));
        flush();   // the header (src/enc.rs:70)
    }
    ~Encoder()   // impl Drop (src/enc.rs:28-34): finish if the caller did not
    {
        if (h_) {
            if (!finished_ && pfv_encoder_finish(h_) == PFV_OK) {
                try { flush(); } catch (...) {}
            }
            pfv_encoder_destroy(h_);
        }
    }
    Encoder(const Encoder &) = delete;
    Encoder &operator=(const Encoder &) = delete;

    void encode_iframe(const VideoFrame &f)   // src/enc.rs:75-123
    {
        check_frame(f);
        ctx_.check(pfv_encoder_encode_iframe(h_, f.plane_y.pixels.data(), f.plane_u.pixels.data(), f.plane_v.pixels.data()));
        flush();
    }
    void encode_pframe(const VideoFrame &f)   // src/enc.rs:125-173
    {
        check_frame(f);
        ctx_.check(pfv_encoder_encode_pframe(h_, f.plane_y.pixels.data(), f.plane_u.pixels.data(), f.plane_v.pixels.data()));
        flush();
    }
    void encode_dropframe()                   // src/enc.rs:175-180
    {
        ctx_.check(pfv_encoder_encode_dropframe(h_));
        flush();
    }
    void finish()                             // src/enc.rs:182-188
    {
        ctx_.check(pfv_encoder_finish(h_));
        finished_ = true;
        flush();
    }
    // packet payloads from the device entropy stage (default) or the host serialisers: same bytes
    void set_device_entropy(bool on) { ctx_.check(pfv_encoder_set_device_entropy(h_, on ? 1 : 0)); }

  private:
    void check_frame(const VideoFrame &f) const   // the asserts of src/enc.rs:76-80
    {
        if (f.width != width_ || f.height != height_ || f.plane_y.pixels.size() != width_ * height_ ||
            f.plane_u.pixels.size() != (width_ / 2) * (height_ / 2) || f.plane_v.pixels.size() != (width_ / 2) * (height_ / 2))
            throw std::invalid_argument("Encoder: frame geometry does not match the encoder (src/enc.rs:76-79)");
    }
    void flush()
    {
        const uint8_t *data = nullptr;
        size_t len = 0;
        ctx_.check(pfv_encoder_drain(h_, &data, &len));   // hands the pending bytes over; the library keeps nothing
        if (len) out_.write(reinterpret_cast<const char *>(data), (std::streamsize)len);
        if (!out_) throw Error(PFV_ERR_IO, "Encoder: the writer failed (the reference propagates io::Error, src/enc.rs:190-235)");
    }
    Context &ctx_;
    std::ostream &out_;
    size_t width_, height_;
    bool finished_ = false;
    pfv_encoder *h_ = nullptr;
};

// src/dec.rs:15-224
class Decoder {
  public:
    using OnVideo = std::function<void(const VideoFrame &)>;

    Decoder(std::istream &reader, Context &ctx)
        : ctx_(ctx), data_((std::istreambuf_iterator<char>(reader)), std::istreambuf_iterator<char>())
    {
        int rc = pfv_decoder_create(ctx.handle(), reinterpret_cast<const uint8_t *>(data_.data()), data_.size(), &h_);
        if (rc != PFV_OK) ctx_.check(rc);   // DecodeError::{FormatError, VersionError, IOError} (src/dec.rs:30-35)
        frame_ = VideoFrame((size_t)width(), (size_t)height());
    }
    ~Decoder() { pfv_decoder_destroy(h_); }
    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;

    uint32_t width() const { return (uint32_t)pfv_decoder_width(h_); }          // src/dec.rs:136-138
    uint32_t height() const { return (uint32_t)pfv_decoder_height(h_); }        // :140-142
    uint32_t framerate() const { return (uint32_t)pfv_decoder_framerate(h_); }  // :144-146
    void reset() { ctx_.check(pfv_decoder_reset(h_)); }                         // :148-152
    void set_lookahead(int n_threads) { ctx_.check(pfv_decoder_set_lookahead(h_, n_threads)); }

    // false at the end of the stream (Ok(false)), true otherwise; onvideo is called for every decoded frame
    bool advance_delta(double delta, const OnVideo &onvideo)                    // src/dec.rs:154-167
    {
        cb_ = &onvideo;
        return result(pfv_decoder_advance_delta(h_, delta, &Decoder::trampoline, this));
    }
    bool advance_frame(const OnVideo &onvideo)                                  // src/dec.rs:169-224
    {
        cb_ = &onvideo;
        return result(pfv_decoder_advance_frame(h_, &Decoder::trampoline, this));
    }

  private:
    bool result(int rc)
    {
        cb_ = nullptr;
        if (rc < 0) ctx_.check(rc);
        return rc == 1;
    }
    static void trampoline(void *user, const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h)
    {
        Decoder *d = static_cast<Decoder *>(user);
        const size_t ny = (size_t)w * h, nc = (size_t)(w / 2) * (h / 2);
        d->frame_.plane_y.pixels.assign(y, y + ny);
        d->frame_.plane_u.pixels.assign(u, u + nc);
        d->frame_.plane_v.pixels.assign(v, v + nc);
        if (d->cb_ && *d->cb_) (*d->cb_)(d->frame_);
    }
    Context &ctx_;
    std::string data_;              // the whole stream: the native decoder reads from it for its lifetime
    pfv_decoder *h_ = nullptr;
    VideoFrame frame_;              // retframe (src/dec.rs:22)
    const OnVideo *cb_ = nullptr;
};

// Encoder with the GOPs of the stream batched per launch (pfv_gop_encoder, pfv_hip.h): same calls, same .pfv bytes; a packet reaches the
// writer when its batch of max_gops groups is complete, on flush() or on finish().  Packets are written segment by segment from where
// they lie (pfv_gop_encoder_drain_iov): no copy on the way to the writer.
class GopEncoder {
  public:
    GopEncoder(std::ostream &writer, size_t width, size_t height, uint32_t framerate, int quality, Context &ctx, int max_gops = 8, int max_gop_frames = 15,
               size_t payload_budget = 0)
        : ctx_(ctx), out_(writer), width_(width), height_(height)
    {
        ctx_.check(pfv_gop_encoder_create(ctx.handle(), (int)width, (int)height, (int)framerate, quality, max_gops, max_gop_frames, payload_budget, &h_));
        drain();   // the header (src/enc.rs:70)
    }
    ~GopEncoder()   // impl Drop (src/enc.rs:28-34)
    {
        if (h_) {
            if (!finished_ && pfv_gop_encoder_finish(h_) == PFV_OK) {
                try { drain(); } catch (...) {}
            }
            pfv_gop_encoder_destroy(h_);
        }
    }
    GopEncoder(const GopEncoder &) = delete;
    GopEncoder &operator=(const GopEncoder &) = delete;
    void encode_iframe(const VideoFrame &f)   // src/enc.rs:75-123
    {
        check_frame(f);
        ctx_.check(pfv_gop_encoder_encode_iframe(h_, f.plane_y.pixels.data(), f.plane_u.pixels.data(), f.plane_v.pixels.data()));
        drain();
    }
    void encode_pframe(const VideoFrame &f)   // src/enc.rs:125-173
    {
        check_frame(f);
        ctx_.check(pfv_gop_encoder_encode_pframe(h_, f.plane_y.pixels.data(), f.plane_u.pixels.data(), f.plane_v.pixels.data()));
        drain();
    }
    // device frames are read where they lie instead of being copied into the batch; the caller keeps each one valid and unchanged until its
    // packet has reached the writer (or flush() / finish() returned)
    void set_frames_by_reference(bool on) { ctx_.check(pfv_gop_encoder_set_frames_by_reference(h_, on ? 1 : 0)); }
    // a packed frame (Y | U | V) that already lies in device memory: nothing crosses PCIe on the way in.  Stream-ordered on the context's
    // stream like every *_dev call: no host wait, the frame may only be overwritten by work enqueued on that stream afterwards
    void encode_iframe_device(const uint8_t *frame_dev)
    {
        ctx_.check(pfv_gop_encoder_encode_iframe_dev(h_, frame_dev));
        drain();
    }
    void encode_pframe_device(const uint8_t *frame_dev)
    {
        ctx_.check(pfv_gop_encoder_encode_pframe_dev(h_, frame_dev));
        drain();
    }
    void encode_dropframe() { ctx_.check(pfv_gop_encoder_encode_dropframe(h_)); }   // src/enc.rs:175-180
    void flush()                              // every frame handed over so far becomes packets at the writer now
    {
        ctx_.check(pfv_gop_encoder_flush(h_));
        drain();
    }
    void finish()                             // src/enc.rs:182-188
    {
        ctx_.check(pfv_gop_encoder_finish(h_));
        finished_ = true;
        drain();
    }
    long batches() const { return pfv_gop_encoder_batches(h_); }

  private:
    void check_frame(const VideoFrame &f) const   // the asserts of src/enc.rs:76-80
    {
        if (f.width != width_ || f.height != height_ || f.plane_y.pixels.size() != width_ * height_ ||
            f.plane_u.pixels.size() != (width_ / 2) * (height_ / 2) || f.plane_v.pixels.size() != (width_ / 2) * (height_ / 2))
            throw std::invalid_argument("GopEncoder: frame geometry does not match the encoder (src/enc.rs:76-79)");
    }
    void drain()
    {
        const pfv_iovec *iov = nullptr;
        size_t n = 0;
        ctx_.check(pfv_gop_encoder_drain_iov(h_, &iov, &n));
        for (size_t i = 0; i < n; i++) out_.write(reinterpret_cast<const char *>(iov[i].data), (std::streamsize)iov[i].len);
        if (!out_) throw Error(PFV_ERR_IO, "GopEncoder: the writer failed (the reference propagates io::Error, src/enc.rs:190-235)");
    }
    Context &ctx_;
    std::ostream &out_;
    size_t width_, height_;
    bool finished_ = false;
    pfv_gop_encoder *h_ = nullptr;
};

// Decoder with the GOPs of the stream batched per launch (pfv_gop_decoder): the frames, their order and the result of every call are
// those of pfv::Decoder; n_threads packet parsers work beside the caller.
class GopDecoder {
  public:
    using OnVideo = std::function<void(const VideoFrame &)>;
    GopDecoder(std::istream &reader, Context &ctx, int max_gops = 8, int max_gop_frames = 15, int n_threads = 4)
        : ctx_(ctx), data_((std::istreambuf_iterator<char>(reader)), std::istreambuf_iterator<char>())
    {
        int rc = pfv_gop_decoder_create(ctx.handle(), reinterpret_cast<const uint8_t *>(data_.data()), data_.size(), max_gops, max_gop_frames, n_threads, &h_);
        if (rc != PFV_OK) ctx_.check(rc);
        frame_ = VideoFrame((size_t)width(), (size_t)height());
    }
    ~GopDecoder() { pfv_gop_decoder_destroy(h_); }
    GopDecoder(const GopDecoder &) = delete;
    GopDecoder &operator=(const GopDecoder &) = delete;
    uint32_t width() const { return (uint32_t)pfv_gop_decoder_width(h_); }
    uint32_t height() const { return (uint32_t)pfv_gop_decoder_height(h_); }
    uint32_t framerate() const { return (uint32_t)pfv_gop_decoder_framerate(h_); }
    void reset() { ctx_.check(pfv_gop_decoder_reset(h_)); }
    // With set_output_device(true) the frames live in device memory: the host-frame callbacks below would read device pointers, so they
    // refuse (use the *_device forms).
    bool advance_delta(double delta, const OnVideo &onvideo)
    {
        if (device_out_) throw std::logic_error("GopDecoder::advance_delta: frames stay in device memory (set_output_device): use advance_delta_device");
        cb_ = &onvideo;
        return result(pfv_gop_decoder_advance_delta(h_, delta, &GopDecoder::trampoline, this));
    }
    bool advance_frame(const OnVideo &onvideo)
    {
        if (device_out_) throw std::logic_error("GopDecoder::advance_frame: frames stay in device memory (set_output_device): use advance_frame_device");
        cb_ = &onvideo;
        return result(pfv_gop_decoder_advance_frame(h_, &GopDecoder::trampoline, this));
    }
    // frames left in device memory (pfv_gop_decoder_set_output_device): the consumer gets the three planes' DEVICE addresses, valid until
    // the call that starts the next batch -- for consumers on the GPU; no frame crosses PCIe
    using OnVideoDevice = std::function<void(const uint8_t *y, const uint8_t *u, const uint8_t *v, uint32_t width, uint32_t height)>;
    void set_output_device(bool on) { ctx_.check(pfv_gop_decoder_set_output_device(h_, on ? 1 : 0)); device_out_ = on; }
    bool advance_frame_device(const OnVideoDevice &onvideo)
    {
        if (!device_out_) throw std::logic_error("GopDecoder::advance_frame_device: set_output_device(true) first");
        dcb_ = &onvideo;
        const int rc = pfv_gop_decoder_advance_frame(h_, &GopDecoder::trampoline_device, this);
        dcb_ = nullptr;
        if (rc < 0) ctx_.check(rc);
        return rc == 1;
    }
    bool advance_delta_device(double delta, const OnVideoDevice &onvideo)       // Decoder::advance_delta (src/dec.rs:154-167), frames in device memory
    {
        if (!device_out_) throw std::logic_error("GopDecoder::advance_delta_device: set_output_device(true) first");
        dcb_ = &onvideo;
        const int rc = pfv_gop_decoder_advance_delta(h_, delta, &GopDecoder::trampoline_device, this);
        dcb_ = nullptr;
        if (rc < 0) ctx_.check(rc);
        return rc == 1;
    }

  private:
    static void trampoline_device(void *user, const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h)
    {
        GopDecoder *d = static_cast<GopDecoder *>(user);
        if (d->dcb_ && *d->dcb_) (*d->dcb_)(y, u, v, (uint32_t)w, (uint32_t)h);
    }
    bool result(int rc)
    {
        cb_ = nullptr;
        if (rc < 0) ctx_.check(rc);
        return rc == 1;
    }
    static void trampoline(void *user, const uint8_t *y, const uint8_t *u, const uint8_t *v, int w, int h)
    {
        GopDecoder *d = static_cast<GopDecoder *>(user);
        const size_t ny = (size_t)w * h, nc = (size_t)(w / 2) * (h / 2);
        d->frame_.plane_y.pixels.assign(y, y + ny);
        d->frame_.plane_u.pixels.assign(u, u + nc);
        d->frame_.plane_v.pixels.assign(v, v + nc);
        if (d->cb_ && *d->cb_) (*d->cb_)(d->frame_);
    }
    Context &ctx_;
    std::string data_;
    pfv_gop_decoder *h_ = nullptr;
    VideoFrame frame_;
    const OnVideo *cb_ = nullptr;
    const OnVideoDevice *dcb_ = nullptr;
    bool device_out_ = false;
};

// n Encoders of one geometry stepped together (pfv_batch_encoder): every writer receives exactly the bytes an Encoder of
// its own would have written (src/enc.rs:12-188); packets arrive one step late, finish() / the destructor flush.
class BatchEncoder {
  public:
    BatchEncoder(std::vector<std::ostream *> writers, size_t width, size_t height, uint32_t framerate, int quality, Context &ctx)
        : ctx_(ctx), writers_(std::move(writers))
    {
        ctx_.check(pfv_batch_encoder_create(ctx.handle(), (int)width, (int)height, (int)framerate, quality, (int)writers_.size(), &on_write, this, &h_));
        frame_bytes_ = pfv_frame_bytes((int)width, (int)height);
    }
    ~BatchEncoder()
    {
        if (h_) {
            if (!finished_) {
                try { finish(); } catch (...) {}
            }
            pfv_batch_encoder_destroy(h_);
        }
    }
    BatchEncoder(const BatchEncoder &) = delete;
    BatchEncoder &operator=(const BatchEncoder &) = delete;
    size_t frame_bytes() const { return frame_bytes_; }
    // page-locked [n][frame_bytes] array to fill for the next encode call (packed Y|U|V per stream)
    uint8_t *frames() { return pfv_batch_encoder_frames(h_); }
    void encode_iframes() { done(pfv_batch_encoder_encode(h_, 0, nullptr)); }
    void encode_pframes() { done(pfv_batch_encoder_encode(h_, 1, nullptr)); }
    void finish()
    {
        const int rc = pfv_batch_encoder_finish(h_);
        finished_ = true;
        done(rc);
    }

  private:
    // A writer that fails (disk full, closed pipe) must not go unnoticed: the reference propagates every write error (`?`,
    // src/enc.rs:190-235).  Nothing may unwind through the C callback, so the failure is remembered, all further writes are
    // dropped (no writer receives bytes behind a hole) and the call that triggered it throws.
    static void on_write(void *user, int stream, const uint8_t *data, size_t len)
    {
        auto *self = static_cast<BatchEncoder *>(user);
        if (self->failed_stream_ >= 0) return;
        std::ostream *w = self->writers_[(size_t)stream];
        w->write(reinterpret_cast<const char *>(data), (std::streamsize)len);
        if (!*w) self->failed_stream_ = stream;
    }
    void done(int rc)
    {
        if (failed_stream_ >= 0) {
            finished_ = true;
            throw Error(PFV_ERR_IO, "BatchEncoder: writer " + std::to_string(failed_stream_) + " failed; its stream is truncated");
        }
        ctx_.check(rc);
    }
    int failed_stream_ = -1;
    Context &ctx_;
    std::vector<std::ostream *> writers_;
    size_t frame_bytes_ = 0;
    bool finished_ = false;
    pfv_batch_encoder *h_ = nullptr;
};

// n Decoders in lockstep (pfv_batch_decoder): advance_frames() -> 1 frames (page-locked [n][frame_bytes], valid until the call
// after next), 2 a step of drop frames, 0 end of the streams; errors as pfv::Error with the DecodeError codes.
class BatchDecoder {
  public:
    BatchDecoder(std::vector<std::string> streams, Context &ctx, int n_threads = 8) : ctx_(ctx), data_(std::move(streams))
    {
        std::vector<const uint8_t *> p;
        std::vector<size_t> n;
        for (const auto &s : data_) { p.push_back(reinterpret_cast<const uint8_t *>(s.data())); n.push_back(s.size()); }
        ctx_.check(pfv_batch_decoder_create(ctx.handle(), p.data(), n.data(), (int)data_.size(), n_threads, &h_));
    }
    ~BatchDecoder() { pfv_batch_decoder_destroy(h_); }
    BatchDecoder(const BatchDecoder &) = delete;
    BatchDecoder &operator=(const BatchDecoder &) = delete;
    size_t width() const { return (size_t)pfv_batch_decoder_width(h_); }
    size_t height() const { return (size_t)pfv_batch_decoder_height(h_); }
    uint32_t framerate() const { return (uint32_t)pfv_batch_decoder_framerate(h_); }
    int advance_frames(const uint8_t **frames)
    {
        const int rc = pfv_batch_decoder_advance(h_, frames);
        if (rc < 0) ctx_.check(rc);
        return rc;
    }

  private:
    Context &ctx_;
    std::vector<std::string> data_;
    pfv_batch_decoder *h_ = nullptr;
};

}  // namespace pfv
