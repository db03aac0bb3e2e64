/*
 * pfv_hip_core.h -- the DROP-IN boundary of libpfv_hip.so, the MI355X (gfx950) implementation of the Pretty Fast Video
 * (pfv-rs 0.2.2, codec 2.1.1) per-macroblock transform / motion path: what a binding of the reference needs, and nothing else.
 *
 *   the six plane operators of `impl VideoPlane` (src/common.rs:351-521)     pfv_encode_plane ... pfv_decode_plane_delta_into
 *   Encoder::new's q-table derivation (src/enc.rs:40-51)                      pfv_qtables_from_quality
 *   the hot-path state of enc::Encoder / dec::Decoder (src/enc.rs:12-26,      pfv_enc_session_*, pfv_enc_iframe / _pframe,
 *     84-97, 134-147; src/dec.rs:15-28, 195-197, 298-323, 419-445)            pfv_dec_session_*, pfv_dec_iframe / _pframe, pfv_dec_get_frame
 *   the stream objects with the reference's own call pattern                  pfv_encoder_*, pfv_decoder_*
 *     (src/enc.rs:37-188, src/dec.rs:38-224)
 *   the `num_threads` slot (src/enc.rs:54, src/dec.rs:125)                    pfv_ctx_create(device)
 *
 * INTEGRATION.md binds exactly this header from Rust (`extern "C"`).  Everything else -- device-pointer forms, batched launch shapes,
 * device entropy stages, diagnostics, the multi-GPU control plane -- is in pfv_hip_ext.h and produces the same bytes, frames and per-call
 * results; pfv_hip.h includes both.
 *
 * Conventions
 *   - Plain pointers and sizes only.  The entry points of this header take HOST pointers, stage through the context and return when
 *     the result is in the caller's buffer (the reference's calls are synchronous: `tp.install` blocks, src/common.rs:374).
 *   - Every function returns PFV_OK (0) or a negative pfv_status; nothing unwinds or aborts across this boundary (the reference
 *     panics on contract violations: src/common.rs:217-218, src/enc.rs:38,76-80).
 *   - Data layouts are the reference's own flattened structs:
 *       coefficients  int16_t[n_mb][4][64]   (EncodedMacroBlock, src/common.rs:9-12: subblocks TL,TR,BL,BR, zigzag order inside)
 *       motion        int8_t [n_mb][2]       (DeltaEncodedMacroBlock.motion_x/_y, :14-19)
 *       has_coef      uint8_t[n_mb]          (subblocks.is_some(); coefficients of a skipped macroblock are written as zeros)
 *       planes        uint8_t row-major, stride = width (VideoPlane, src/plane.rs:1-5)
 *     Macroblocks are in raster order (src/common.rs:364-369); planes in Y,U,V order.
 *   - Quantiser tables are int32_t[64] in raster order with every entry in [1, 65535] (the file format stores them as u16,
 *     src/enc.rs:201-215).
 *   - A context is not thread-safe; distinct contexts are independent (one HIP stream each), like distinct Encoder / Decoder
 *     instances with their own rayon pools.  There is no CPU fallback: without a gfx950 device pfv_ctx_create returns
 *     PFV_ERR_NO_DEVICE.
 */
#ifndef PFV_HIP_CORE_H
#define PFV_HIP_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFV_API __attribute__((visibility("default")))

typedef enum pfv_status {
    PFV_OK = 0,
    PFV_ERR_BAD_ARG = -1,   /* null pointer, non-positive size, q entry outside [1,65535], quality outside 0..10 */
    PFV_ERR_HIP = -2,       /* a HIP runtime call failed; see pfv_last_error() */
    PFV_ERR_NOMEM = -3,
    PFV_ERR_BAD_MV = -4,    /* a motion vector points outside the reference plane (src/common.rs:258-259) */
    PFV_ERR_NO_DEVICE = -5, /* no usable gfx950 device */
    PFV_ERR_FORMAT = -6,    /* DecodeError::FormatError (src/dec.rs:30-35) */
    PFV_ERR_VERSION = -7,   /* DecodeError::VersionError */
    PFV_ERR_IO = -8,        /* truncated / unreadable stream */
    PFV_ERR_STATE = -9      /* e.g. encode after finish (src/enc.rs:80 assert) */
} pfv_status;

typedef struct pfv_ctx pfv_ctx;

/* ------------------------------------------------------------------ context */
/* Replaces the `num_threads` / rayon::ThreadPool slot of Encoder::new (src/enc.rs:37,54)
 * and Decoder::new (src/dec.rs:38,125): the parallel resource is a device + stream. */
/* number of HIP devices visible to the process (0 when there is none) */
PFV_API int pfv_device_count(void);
PFV_API int pfv_ctx_create(int device, pfv_ctx **out);
/* also tears down what was left alive on the context (pfv_hip_ext.h: communicators; private contexts of GOP encoders are detached) */
PFV_API void pfv_ctx_destroy(pfv_ctx *ctx);
PFV_API int pfv_ctx_sync(pfv_ctx *ctx);
/* last error text of this context (or of the calling thread when ctx == NULL) */
PFV_API const char *pfv_last_error(pfv_ctx *ctx);
PFV_API const char *pfv_version(void);

/* x + (16 - x%16)%16  (src/common.rs:352-353, src/frame.rs:29-36) */
PFV_API int pfv_pad16(int x);

/* Encoder::new q-table derivation (src/enc.rs:40-51).  quality in 0..10. */
PFV_API int pfv_qtables_from_quality(int quality, int32_t intra_l[64], int32_t intra_c[64], int32_t inter_l[64],
                                     int32_t inter_c[64], float *px_err);

/* ------------------------------------------------------------------ the six plane-level operators, host buffers */
/* VideoPlane::encode_plane (src/common.rs:351-386).
 * px: w*h source plane.  coef_out: pad16(w)/16 * pad16(h)/16 macroblocks * 256 int16. */
PFV_API int pfv_encode_plane(pfv_ctx *ctx, const uint8_t *px, int w, int h, const int32_t q[64], uint8_t clear,
                             int16_t *coef_out);

/* VideoPlane::encode_plane_delta (src/common.rs:388-421).
 * ref: previous reconstructed plane, pad16(w) x pad16(h). */
PFV_API int pfv_encode_plane_delta(pfv_ctx *ctx, const uint8_t *px, int w, int h, const uint8_t *ref,
                                   const int32_t q[64], float px_err, uint8_t clear, int8_t *mv_out,
                                   uint8_t *has_coef_out, int16_t *coef_out);

/* VideoPlane::decode_plane (src/common.rs:423-446) and decode_plane_into (:477-496):
 * target is the bw*16 x bh*16 plane; every pixel is overwritten. */
PFV_API int pfv_decode_plane_into(pfv_ctx *ctx, const int16_t *coef, int bw, int bh, const int32_t q[64],
                                  uint8_t *target);

/* VideoPlane::decode_plane_delta (src/common.rs:448-475): ref -> out (distinct buffers). */
PFV_API int pfv_decode_plane_delta(pfv_ctx *ctx, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef,
                                   int bw, int bh, const int32_t q[64], const uint8_t *ref, uint8_t *out);

/* VideoPlane::decode_plane_delta_into (src/common.rs:498-521): read-all-then-write-all
 * into the same plane. */
PFV_API int pfv_decode_plane_delta_into(pfv_ctx *ctx, const int8_t *mv, const uint8_t *has_coef,
                                        const int16_t *coef, int bw, int bh, const int32_t q[64],
                                        uint8_t *ref_and_target);

/* ------------------------------------------------------------------ encoder session (hot-path half of enc::Encoder)
 * Holds what Encoder holds for the hot path (src/enc.rs:12-26): width/height, the four
 * q-tables, px_err and `prev_frame` (padded, resident in HBM, ping-ponged), for
 * `n_streams` independent streams processed in one launch per frame step.
 *
 * Frame layout handed to the session ("frame" = VideoFrame, src/frame.rs:3-9):
 *   one stream's frame = Y (w*h) | U (w/2*h/2) | V (w/2*h/2), tightly packed;
 *   n_streams frames back to back.  pfv_frame_bytes(w,h) gives the size of one.
 * Outputs per stream: total_blocks = blocks(Y)+blocks(U)+blocks(V) macroblocks in
 * Y,U,V order -- exactly the order write_iframe_packet / write_pframe_packet consume
 * (src/enc.rs:247-287, 342-400, 414-451). */
typedef struct pfv_enc_session pfv_enc_session;

PFV_API size_t pfv_frame_bytes(int width, int height);
PFV_API size_t pfv_padded_frame_bytes(int width, int height);
PFV_API int pfv_total_blocks(int width, int height);

/* The encode kernels run the transforms of the closed loop in f32 where that is provably the same arithmetic (every
 * intermediate an integer below 2^24 for the session's tables -- checked here at creation; always true for quality 0..10) and
 * in i32 otherwise (pfv_hip_ext.h: PFV_OPT_ENC_TRANSFORM forces the integer kernels, a diagnostic).  Same bytes. */
PFV_API int pfv_enc_session_create(pfv_ctx *ctx, int width, int height, int quality, int n_streams,
                                   pfv_enc_session **out);
PFV_API void pfv_enc_session_destroy(pfv_enc_session *s);
/* host-buffer forms (one call = upload + launch + download + sync) */
PFV_API int pfv_enc_iframe(pfv_enc_session *s, const uint8_t *frames, int16_t *coef_out);
PFV_API int pfv_enc_pframe(pfv_enc_session *s, const uint8_t *frames, int8_t *mv_out, uint8_t *has_coef_out,
                           int16_t *coef_out);
/* copy prev_frame of all streams (padded) to host: n_streams * pfv_padded_frame_bytes */
PFV_API int pfv_enc_prev_frame(pfv_enc_session *s, uint8_t *out_host);

/* ------------------------------------------------------------------ decoder session (hot-path half of dec::Decoder)
 * Holds `qtables` and the padded `framebuffer` (src/dec.rs:15-28), n_streams-wide.
 * qtables: n_qtables tables of 64 (header order: intra_l, intra_c, inter_l, inter_c;
 * src/enc.rs:199-215). */
typedef struct pfv_dec_session pfv_dec_session;

PFV_API int pfv_dec_session_create(pfv_ctx *ctx, int width, int height, const int32_t *qtables, int n_qtables,
                                   int n_streams, pfv_dec_session **out);
PFV_API void pfv_dec_session_destroy(pfv_dec_session *s);
PFV_API int pfv_dec_iframe(pfv_dec_session *s, const int16_t *coef, const uint8_t qidx[3]);
PFV_API int pfv_dec_pframe(pfv_dec_session *s, const int8_t *mv, const uint8_t *has_coef, const int16_t *coef,
                           const uint8_t qidx[3]);
PFV_API int pfv_dec_get_frame(pfv_dec_session *s, uint8_t *frames_out);
/* padded framebuffer of all streams to host */
PFV_API int pfv_dec_framebuffer(pfv_dec_session *s, uint8_t *out_host);
/* bad-motion-vector flag (src/common.rs:258-259) raised by the last p-frame decode(s): PFV_ERR_BAD_MV; reading it syncs.  The host-buffer forms check it
 * themselves. */
PFV_API int pfv_dec_check(pfv_dec_session *s);

/* ------------------------------------------------------------------ stream-level session objects: enc::Encoder / dec::Decoder
 * enc::Encoder<W: Write> (src/enc.rs:12-188) with the writer being an in-memory byte vector, and
 * dec::Decoder<R: Read + Seek> (src/dec.rs:15-224) over a caller-owned byte slice.  The per-macroblock work runs
 * on the device sessions above; RLE / Huffman / bit packing and the container run on the host.
 * Planes: y = width*height, u and v = (width/2)*(height/2) (VideoFrame, src/frame.rs:3-9). */
typedef struct pfv_encoder pfv_encoder;
typedef struct pfv_decoder pfv_decoder;
typedef void (*pfv_video_cb)(void *user, const uint8_t *y, const uint8_t *u, const uint8_t *v, int width, int height);

PFV_API int pfv_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, pfv_encoder **out);
PFV_API int pfv_encoder_encode_iframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* PFV_ERR_STATE when the previous frame failed after prev_frame had advanced (oversize coefficient, out of memory, HIP
 * error while fetching the payload): the stream no longer matches the encoder's reference; an i-frame clears this. */
PFV_API int pfv_encoder_encode_pframe(pfv_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v);
PFV_API int pfv_encoder_encode_dropframe(pfv_encoder *e);
PFV_API int pfv_encoder_finish(pfv_encoder *e);
/* Writer side (the reference's `W: Write`, src/enc.rs:12-26): pfv_encoder_drain hands over the bytes produced since the last
 * drain (header after create, one packet per encode call) and forgets them -- valid until the next call on this encoder;
 * nothing accumulates in the library.  pfv_encoder_bytes peeks at the bytes not yet drained without consuming them. */
PFV_API int pfv_encoder_drain(pfv_encoder *e, const uint8_t **data, size_t *len);
PFV_API int pfv_encoder_bytes(pfv_encoder *e, const uint8_t **data, size_t *len);
PFV_API void pfv_encoder_destroy(pfv_encoder *e);

/* `data` must stay valid while the decoder lives.  Errors: PFV_ERR_FORMAT / PFV_ERR_VERSION / PFV_ERR_IO
 * = DecodeError::{FormatError, VersionError, IOError} (src/dec.rs:30-35). */
PFV_API int pfv_decoder_create(pfv_ctx *ctx, const uint8_t *data, size_t len, pfv_decoder **out);
PFV_API void pfv_decoder_destroy(pfv_decoder *d);
PFV_API int pfv_decoder_width(const pfv_decoder *d);
PFV_API int pfv_decoder_height(const pfv_decoder *d);
PFV_API int pfv_decoder_framerate(const pfv_decoder *d);
PFV_API int pfv_decoder_reset(pfv_decoder *d);
/* 1 = Ok(true) (more data), 0 = Ok(false) (EOF), negative = error */
PFV_API int pfv_decoder_advance_frame(pfv_decoder *d, pfv_video_cb onvideo, void *user);
PFV_API int pfv_decoder_advance_delta(pfv_decoder *d, double delta, pfv_video_cb onvideo, void *user);

#ifdef __cplusplus
}
#endif
#endif /* PFV_HIP_CORE_H */
