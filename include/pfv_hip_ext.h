/*
 * pfv_hip_ext.h -- everything of libpfv_hip.so's C ABI that is NOT the drop-in boundary (that is pfv_hip_core.h, included here).
 * Same conventions; `*_dev` entry points take DEVICE pointers and are asynchronous on the context's HIP stream.
 *
 *   [B] THROUGHPUT FORMS of the core's operations -- same bytes, frames and per-call results, another launch shape or residence;
 *       each is parity-tested against the core and the oracle: the *_dev entry points and device memory helpers, HIP graphs, the
 *       session window / frame stride (GOP batching), the device entropy stage of the encoder session (pfv_enc_entropy_*,
 *       pfv_enc_pack_*), sparse pairs and coefficient lists into the decoder session, the batch objects (pfv_batch_*), the GOP-batched
 *       objects (pfv_gop_*), frames left in / taken from device memory, look-ahead threads of pfv_decoder, plane helpers on device planes
 *       (blit, reduce, double, RGB <-> YUV).
 *   [C] DIAGNOSTICS, MEASUREMENT AND TEST HOOKS -- not needed by a caller of the codec and free to change: context options
 *       (pfv_ctx_set_option), events, the synthetic workload generator, the payload serialisers / parsers on their own, object
 *       statistics, and the multi-GPU control plane (pfv_comm_*), which serves bench.py's sharded job, not the codec.
 */
#ifndef PFV_HIP_EXT_H
#define PFV_HIP_EXT_H

#include "pfv_hip_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same with a priority for the context's HIP stream: > 0 the device's greatest, < 0 its least, 0 the default.  When an
 * encoder and a decoder work side by side on two contexts (a transcoder; bench.py's single-stream schedule), the encoder's launches
 * are the critical path and the decoder's fill the gaps: encoder context high, decoder context low. */
PFV_API int pfv_ctx_create_prio(int device, int priority, pfv_ctx **out);
/* PCI address "domain:bus:device.function" of the context's device (len >= 16): which physical GPU a rank sits on */
PFV_API int pfv_ctx_pci_bus_id(pfv_ctx *ctx, char *out, int len);
/* hipDeviceSynchronize on the context's device (all streams) */
PFV_API int pfv_device_sync(pfv_ctx *ctx);
/* hipStream_t of the context (for callers that enqueue their own work / HIP events) */
PFV_API void *pfv_ctx_stream(pfv_ctx *ctx);

/* [C] Context options (diagnostics / test parametrisation; the defaults are what production wants).  An option applies to the
 * plane-level operators called on the context and to sessions CREATED afterwards (a session keeps the values it was created
 * with).  Results are the same bytes under every value.
 *   PFV_OPT_ENC_TRANSFORM  how the encode kernels evaluate the transforms of the closed loop (src/dct.rs:176-293):
 *       PFV_ENC_TRANSFORM_AUTO (default)  in f32 where that is provably the same arithmetic -- every intermediate an integer
 *                                          below 2^24 for the session's tables, checked at session creation; always true for
 *                                          quality 0..10 -- and in i32 otherwise
 *       PFV_ENC_TRANSFORM_INT             always the i32 kernels
 *   PFV_OPT_TILE_COMPACTION  1 (default): the p-frame encoder transforms only CODED macroblocks where that saves work -- a
 *       skipped macroblock is not transformed by the reference either (src/common.rs:221-222) -- by moving the coded macroblocks of
 *       a 128 x 64 tile together before the transform phase; 0: every wavefront transforms its own strip (measurements);
 *       2: the p-frame encoder as TWO kernels (measurements: 17 % slower, profiles/r06_enc_pframe_split.md) -- k_pf_search (search, skip
 *       decision, skipped macroblocks finished) and k_pf_transform (coded macroblocks only, numbered per 64 macroblocks); launches of
 *       the 8-lanes-per-macroblock mapping only, the small-grid mapping ignores it
 *   PFV_OPT_LANE_MAPPING  how the four codec kernels spread a macroblock over lanes:
 *       PFV_LANES_AUTO (default)   by grid size: 8 lanes per macroblock (a wavefront = a strip of 8 macroblocks) for launches that
 *                                  fill the device, 16 (a wavefront = 4 macroblocks, half as long) for launches of fewer than
 *                                  4 096 strips -- one or two 1080p streams per launch, the reference's own usage
 *                                  (src/enc.rs:125-173)
 *       PFV_LANES_PER_MB_8 / PFV_LANES_PER_MB_16   force one of them
 *   PFV_OPT_ENTROPY_DECODE  where pfv_gop_decoder turns packet payloads into coefficients (src/dec.rs:258-296, 378-417):
 *       PFV_ENTROPY_DECODE_AUTO (default)  on the device (k_entd_*: self-synchronising parallel read of the run streams) when the
 *                                          batch's coefficient lists fit the device's free memory, on the host otherwise
 *       PFV_ENTROPY_DECODE_HOST            the host parser pool (n_threads of pfv_gop_decoder_create)
 *       PFV_ENTROPY_DECODE_DEVICE          the device, or PFV_ERR_NOMEM from pfv_gop_decoder_create
 *     Either way a payload the device stage is not sure about (damaged, degenerate code table, periodic content whose read does
 *     not settle) is parsed by the host code, which alone decides about errors.
 *   PFV_OPT_ENTDEC_LANE_BITS / _LAUNCHES / _INNER_ROUNDS  shape of the device stage (measurements, and tests that force the "not settled"
 *       road): payload bits per lane (a multiple of 32 in 32..256, default 256), read launches before the verifying one (1..64, default 3:
 *       the full read k_entd_sync, then k_entd_fix for the seams between its workgroups), settling rounds inside a workgroup of the full
 *       read (1..1024, default 96; a round in which no lane has a new start ends them) */
typedef enum pfv_option {
    PFV_OPT_ENC_TRANSFORM = 1, PFV_OPT_TILE_COMPACTION = 2, PFV_OPT_LANE_MAPPING = 3, PFV_OPT_ENTROPY_DECODE = 4,
    PFV_OPT_ENTDEC_LANE_BITS = 5, PFV_OPT_ENTDEC_LAUNCHES = 6, PFV_OPT_ENTDEC_INNER_ROUNDS = 7
} pfv_option;
enum { PFV_ENTROPY_DECODE_AUTO = 0, PFV_ENTROPY_DECODE_HOST = 1, PFV_ENTROPY_DECODE_DEVICE = 2 };
enum { PFV_LANES_AUTO = 0, PFV_LANES_PER_MB_8 = 1, PFV_LANES_PER_MB_16 = 2 };
enum { PFV_ENC_TRANSFORM_AUTO = 0, PFV_ENC_TRANSFORM_INT = 1 };
PFV_API int pfv_ctx_set_option(pfv_ctx *ctx, int option, int value);
PFV_API int pfv_ctx_get_option(pfv_ctx *ctx, int option, int *value);

/* [C] Timing events on the context's stream (HIP events): record costs a microsecond or two, so every launch of a pass can be
 * bracketed without disturbing it; pfv_event_elapsed_ms waits for the later event. */
typedef struct pfv_event pfv_event;
PFV_API int pfv_event_create(pfv_ctx *ctx, pfv_event **out);
PFV_API int pfv_event_record(pfv_event *e);
PFV_API int pfv_event_elapsed_ms(pfv_event *start, pfv_event *stop, float *ms);
PFV_API void pfv_event_destroy(pfv_event *e);
/* [B] Ordering between contexts (each has its own HIP stream): ctx's stream waits, on the device, for an event recorded on another
 * context's stream.  This is how a decoder on one context consumes what an encoder on another produces while the encoder is
 * already working on the next frame -- Encoder and Decoder are independent objects in the reference (src/enc.rs:12-26,
 * src/dec.rs:15-28), and for a single stream the device is far from full with one of them. */
PFV_API int pfv_ctx_wait_event(pfv_ctx *ctx, pfv_event *e);

/* [B] HIP graphs over the `*_dev` entry points.  The reference's caller is one Encoder per stream, one call per frame
 * (src/enc.rs:125-173); for a single stream the launches, not the kernels, are the cost.  Every `*_dev` call made between
 * pfv_graph_begin and pfv_graph_end on this context is recorded instead of executed (stream capture); pfv_graph_launch
 * replays the whole sequence -- e.g. the 30 launches of a GOP-15 encode + decode -- as one launch.  Device pointers are baked
 * into the graph.  The recorded sequence must start with an i-frame step of every session it touches (an i-frame reads no
 * previous state, src/enc.rs:84-97), or contain an even number of frame steps per session, so that the sessions' ping-pong
 * state after a replay equals the state after the recording.  Host-pointer entry points cannot be recorded. */
typedef struct pfv_graph pfv_graph;
PFV_API int pfv_graph_begin(pfv_ctx *ctx);
PFV_API int pfv_graph_end(pfv_ctx *ctx, pfv_graph **out);
PFV_API int pfv_graph_launch(pfv_graph *g);
PFV_API void pfv_graph_destroy(pfv_graph *g);

/* VideoPlane::blit (src/plane.rs:20-29) on device-resident planes. */
PFV_API int pfv_blit_dev(pfv_ctx *ctx, uint8_t *dst, int dst_w, int dst_h, const uint8_t *src, int src_w, int src_h,
                         int dx, int dy, int sx, int sy, int sw, int sh);

/* VideoPlane::reduce (src/common.rs:523-536, point-sampled 2x decimation, dst = src_w/2 x src_h/2) and
 * VideoPlane::double (:538-556, nearest 2x upsampling, dst = 2 src_w x 2 src_h) on device-resident planes. */
PFV_API int pfv_reduce_dev(pfv_ctx *ctx, uint8_t *dst, const uint8_t *src, int src_w, int src_h);
PFV_API int pfv_double_dev(pfv_ctx *ctx, uint8_t *dst, const uint8_t *src, int src_w, int src_h);
/* The RGB <-> YCbCr helpers of the reference's tests (src/lib.rs:337-394; JPEG-conversion matrix in f32, `as u8`):
 * interleaved RGB8 (width*height*3) -> packed Y|U|V 4:2:0 frame as load_frame + VideoFrame::from_planes produce it
 * (chroma point-sampled at even pixels, src/frame.rs:51-59), and back as save_frame does (chroma doubled,
 * src/common.rs:538-556).  width, height even.  Device-resident buffers. */
PFV_API int pfv_rgb_to_yuv420_dev(pfv_ctx *ctx, const uint8_t *rgb_dev, int width, int height, uint8_t *frame_dev);
PFV_API int pfv_yuv420_to_rgb_dev(pfv_ctx *ctx, const uint8_t *frame_dev, int width, int height, uint8_t *rgb_dev);

/* ------------------------------------------------------------------ multi-GPU control plane (one process per GPU, RCCL over xGMI)  [C]
 * The path shards by stream and by GOP (src/enc.rs:12-26, 84-97): no data-path collective exists.  These carry the few hundred
 * bytes that do travel -- the assignment table (broadcast) and the per-rank counters (reduction / gather) -- on the context's
 * HIP stream.  Rank 0 creates the id, every rank of the job gets the same 128 bytes over the launcher's own channel
 * (pretty-fast-video_amd/comm.py: TCP on MASTER_ADDR) and calls pfv_comm_init on the context of ITS device.  librccl.so is
 * opened at run time; PFV_ERR_NO_DEVICE when it is missing.  world = 1 is legal (a 1-rank communicator).  A communicator
 * belongs to its context (it enqueues on the context's stream): destroy it first, or leave it to pfv_ctx_destroy, which tears down
 * the communicators still alive.  PFV_ERR_STATE while the context records a graph (pfv_graph_begin). */
typedef struct pfv_comm pfv_comm;
enum { PFV_COMM_SUM = 0, PFV_COMM_MAX = 1 };
PFV_API int pfv_comm_unique_id(uint8_t id_out[128]);
PFV_API int pfv_comm_init(pfv_ctx *ctx, int rank, int world, const uint8_t id[128], pfv_comm **out);
PFV_API int pfv_comm_rank(const pfv_comm *c);
PFV_API int pfv_comm_world(const pfv_comm *c);
/* in-place collectives on DEVICE buffers, asynchronous on the context's stream */
PFV_API int pfv_comm_broadcast_dev(pfv_comm *c, void *buf_dev, size_t bytes, int root);
PFV_API int pfv_comm_allreduce_f64_dev(pfv_comm *c, double *buf_dev, size_t count, int op);
PFV_API int pfv_comm_allgather_dev(pfv_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
/* host values (count <= 64): staged through the device, reduced on the stream, synchronised */
PFV_API int pfv_comm_allreduce_f64(pfv_comm *c, double *values, size_t count, int op);
/* all ranks have arrived and everything enqueued before on their streams is done */
PFV_API int pfv_comm_barrier(pfv_comm *c);
/* a communicator belongs to its context: pfv_ctx_destroy tears down the ones still alive, and their handles are invalid from then on */
PFV_API void pfv_comm_destroy(pfv_comm *c);

/* ------------------------------------------------------------------ synthetic workload (not a reference interface)  [C]
 * The reference's fixtures are Git-LFS stubs; tests and benchmarks run on an integer-only synthetic video (SURVEY.md
 * section 8d) that is generated where it is consumed: frame `t` of n_streams streams (stream s seeded with seeds[s], a HOST
 * array) as packed Y|U|V frames back to back in frames_dev.  Byte-identical to synth.SyntheticStream(w, h, seed).frame(t). */
PFV_API int pfv_synth_frames_dev(pfv_ctx *ctx, int width, int height, int n_streams, const uint64_t *seeds, int t,
                                 uint8_t *frames_dev);
/* The same with the content kind chosen: PFV_SYNTH_PAN (what pfv_synth_frames_dev generates: the whole texture pans, noise on
 * half the macroblocks: ~85 % of a quality-5 p-frame is coded) or PFV_SYNTH_LOW_MOTION (static background, four noisy rectangles
 * of about a quarter of the frame's width and height moving over it: ~25 % coded, the rest skipped, src/common.rs:221-222).
 * PFV_SYNTH_STATIC: the background alone (every p-frame macroblock skipped: the floor of the p-frame encoder, its search).
 * Byte-identical to synth.SyntheticStream(w, h, seed, kind).frame(t). */
enum { PFV_SYNTH_PAN = 0, PFV_SYNTH_LOW_MOTION = 1, PFV_SYNTH_STATIC = 2 };
PFV_API int pfv_synth_frames_kind_dev(pfv_ctx *ctx, int width, int height, int n_streams, const uint64_t *seeds, int t, int kind,
                                      uint8_t *frames_dev);

/* ------------------------------------------------------------------ device memory helpers  [B] */
PFV_API int pfv_dev_alloc(pfv_ctx *ctx, size_t bytes, void **out);
PFV_API int pfv_dev_free(pfv_ctx *ctx, void *p);
/* device-to-device copy, asynchronous on the context's stream (ordered like every *_dev call) */
PFV_API int pfv_dev_copy(pfv_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
/* page-locked host memory (hipHostMalloc) for the buffers handed to the host-pointer entry points: uploads / downloads
 * from it run at PCIe rate instead of bouncing through the runtime's staging (the reference's Vec<u8> planes,
 * src/plane.rs:1-5, would be allocated here by a binding that cares) */
PFV_API int pfv_host_alloc(pfv_ctx *ctx, size_t bytes, void **out);
PFV_API int pfv_host_free(pfv_ctx *ctx, void *p);
PFV_API int pfv_dev_upload(pfv_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
PFV_API int pfv_dev_download(pfv_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* Encoder::encode_iframe, hot-path part (src/enc.rs:84-97): per plane encode_plane ->
 * decode_plane -> prev_frame.blit, fused into one launch over all streams and planes. */
PFV_API int pfv_enc_iframe_dev(pfv_enc_session *s, const uint8_t *frames_dev, int16_t *coef_dev);
/* Encoder::encode_pframe, hot-path part (src/enc.rs:134-147). */
PFV_API int pfv_enc_pframe_dev(pfv_enc_session *s, const uint8_t *frames_dev, int8_t *mv_dev, uint8_t *has_coef_dev,
                               int16_t *coef_dev);
/* GOP-batched use of a session.  The n_streams slots of a session need not be different videos: encode_iframe never reads
 * prev_frame and overwrites every plane of it (src/enc.rs:84-97), so the GOPs of ONE stream are independent of each other and
 * can occupy the slots -- frame t of every GOP in one launch (bench.py --workload config5; pfv_gop_encoder below).
 *   pfv_enc_session_set_frame_stride: bytes between the input frames of consecutive slots (0 = packed, the default).  With
 *       the stream's frames resident in display order and equal GOPs of G frames, stride = G * pfv_frame_bytes and
 *       frames_dev = first frame + t * pfv_frame_bytes make slot g read frame g * G + t.
 *   pfv_enc_session_set_window: the following pfv_enc_*frame_dev / pfv_enc_pack_*_dev calls work on slots
 *       [first, first + count) only (a shorter last GOP); the buffers keep their full-width layout, entries of other slots are
 *       left alone.  A slot left out of a frame step keeps no usable prev_frame: its next frame must be an i-frame.
 * The host-buffer entry points need the full window and packed frames (PFV_ERR_STATE otherwise). */
PFV_API int pfv_enc_session_set_frame_stride(pfv_enc_session *s, size_t stride_bytes);
PFV_API int pfv_enc_session_set_window(pfv_enc_session *s, int first, int count);
/* device pointer of stream `stream`'s current prev_frame (padded Y|U|V), for checks */
PFV_API const uint8_t *pfv_enc_prev_frame_dev(pfv_enc_session *s, int stream);

/* ------------------------------------------------------------------ device entropy stage (encoder session)  [B]
 * The reference serialises each frame on one host thread: rle_encode per macroblock (src/rle.rs:9-47), one histogram
 * and Huffman tree per frame (rle.rs:40-66, src/huffman.rs:71-119), LSB-first bit packing into the packet payload
 * (write_iframe_packet src/enc.rs:237-320, write_pframe_packet :332-470).  These entry points build the same payload
 * bytes on the device from the buffers pfv_enc_iframe_dev / pfv_enc_pframe_dev produced, so only the compressed
 * payload crosses PCIe.  Payloads are byte-identical to pfv_serialize_iframe_payload / pfv_serialize_pframe_payload. */
/* upper bound of a payload for this geometry, in bytes (multiple of 4) */
PFV_API size_t pfv_payload_worst_case(int width, int height);
/* allocates the stage's buffers; payload_cap = bytes per stream (0: pfv_payload_worst_case).  Idempotent. */
PFV_API int pfv_enc_entropy_enable(pfv_enc_session *s, size_t payload_cap);
/* 1: the stage runs on its own HIP stream, so the (memory-bound) entropy kernels of frame t overlap the (VALU-bound)
 * encode kernel of frame t+1.  The caller must then alternate between TWO sets of device buffers for the encode outputs
 * it packs.  pfv_enc_payload_sizes / _fetch synchronise with the stage; pfv_enc_entropy_join makes the context's stream
 * wait for it without blocking the host.  0 (default): everything on the context's stream. */
PFV_API int pfv_enc_entropy_set_async(pfv_enc_session *s, int on);
PFV_API int pfv_enc_entropy_join(pfv_enc_session *s);
/* coef_dev / mv_dev / has_coef_dev: device buffers in the layout the encode entry points write (n_streams wide) */
PFV_API int pfv_enc_pack_iframe_dev(pfv_enc_session *s, const int16_t *coef_dev);
PFV_API int pfv_enc_pack_pframe_dev(pfv_enc_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev,
                                    const int16_t *coef_dev);
/* byte count of each stream's payload from the last pack call (synchronises).  PFV_ERR_FORMAT: a coefficient needs more
 * than 15 size bits (the reference panics, rle.rs:44); PFV_ERR_NOMEM: a payload exceeds the capacity.  sizes_out is
 * filled either way (0 for failed streams). */
PFV_API int pfv_enc_payload_sizes(pfv_enc_session *s, uint32_t *sizes_out);
/* every stream's payload with one device-to-host copy: gathered back to back on the device (16-byte aligned starts) into
 * `out` (cap bytes, ideally from pfv_host_alloc); stream s occupies out[offsets_out[s] .. + sizes_out[s]).  Synchronises;
 * errors as pfv_enc_payload_sizes, PFV_ERR_NOMEM when cap is too small. */
PFV_API int pfv_enc_payloads_fetch(pfv_enc_session *s, uint8_t *out, size_t cap, uint32_t *sizes_out, uint64_t *offsets_out);
PFV_API const uint8_t *pfv_enc_payload_dev(pfv_enc_session *s, int stream);
PFV_API size_t pfv_enc_payload_capacity(pfv_enc_session *s);
/* first nbytes of one stream's payload to the host (synchronises) */
PFV_API int pfv_enc_payload_fetch(pfv_enc_session *s, int stream, uint8_t *out_host, size_t nbytes);
/* Decoder::decode_iframe after entropy decoding (src/dec.rs:298-323 -> deserialize_plane
 * :450-479 -> decode_plane_into).  qidx: the three per-plane q-table indices of the
 * packet (src/dec.rs:249-251). */
PFV_API int pfv_dec_iframe_dev(pfv_dec_session *s, const int16_t *coef_dev, const uint8_t qidx[3]);
/* Decoder::decode_pframe after entropy decoding (src/dec.rs:419-445 -> :481-517). */
PFV_API int pfv_dec_pframe_dev(pfv_dec_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev,
                               const int16_t *coef_dev, const uint8_t qidx[3]);
/* Sparse forms: the non-zero coefficients as n (flat index into [stream][macroblock][256], value) pairs -- what the
 * bit parser (src/dec.rs:261-296, 378-417) produces before it is spread into the dense vector; ~10x fewer bytes over
 * PCIe.  Same result as the dense call on the expanded array; indices past the frame are ignored. */
PFV_API int pfv_dec_iframe_sparse(pfv_dec_session *s, const uint32_t *idx, const int16_t *val, size_t n, const uint8_t qidx[3]);
PFV_API int pfv_dec_pframe_sparse(pfv_dec_session *s, const int8_t *mv, const uint8_t *has_coef, const uint32_t *idx,
                                  const int16_t *val, size_t n, const uint8_t qidx[3]);
/* Coefficient lists (round 5): a frame's non-zero coefficients as 32-bit entries instead of its dense [macroblock][256] array -- the form the
 * stream decoders' entropy stage hands to the decode kernels, which expand a strip's entries straight into their LDS zigzag stage (the
 * reference expands runs into the macroblock it is about to decode, src/dec.rs:258-296, 378-417); nothing is cleared and nothing but the
 * values travels.
 *   entry   value (i16) << 16 | (macroblock index & 255) << 8 | position in the macroblock (0..255), ascending by (macroblock, position);
 *   count   per macroblock m and one more behind the last (total_blocks + 1 per frame): the entries that belong to macroblocks before m,
 *           so m owns [count[m], count[m + 1]); a macroblock a p-frame skips owns none.  The counts are the kernels' loop bounds: they
 *           are not validated on the device.
 * entries_dev: per slot of the session's window a DEVICE pointer to the slot's list (a device array of device pointers);
 * counts_dev: [slot][total_blocks + 1].  Same result as the dense call on the expanded arrays. */
PFV_API int pfv_dec_iframe_lists_dev(pfv_dec_session *s, const uint32_t *const *entries_dev, const uint32_t *counts_dev, const uint8_t qidx[3]);
PFV_API int pfv_dec_pframe_lists_dev(pfv_dec_session *s, const int8_t *mv_dev, const uint8_t *has_coef_dev, const uint32_t *const *entries_dev,
                                     const uint32_t *counts_dev, const uint8_t qidx[3]);
/* host helper: one frame's dense coefficients ([total_blocks][256]; has_coef NULL: every macroblock is read) as a coefficient list.  Room for
 * `cap` entries and total_blocks + 1 counts; *n_out = entries written; returns 1 when `cap` does not suffice (total_blocks x 256 always does). */
PFV_API int pfv_coef_lists_from_dense(const int16_t *coef, const uint8_t *has_coef, int total_blocks, uint32_t *entries_out, size_t cap,
                                      uint32_t *counts_out, size_t *n_out);
/* Decoder::advance_frame's crop of framebuffer into retframe (src/dec.rs:195-197,
 * 209-211): frames_out = n_streams unpadded frames (Y|U|V). */
PFV_API int pfv_dec_get_frame_dev(pfv_dec_session *s, uint8_t *frames_out_dev);
/* Fused form of the same crop: once a device buffer of n_streams unpadded frames is set, every
 * following pfv_dec_iframe_dev / pfv_dec_pframe_dev also writes the retframe into it (the decode
 * kernels store each reconstructed row twice: padded framebuffer + cropped retframe), saving the
 * separate blit pass.  NULL switches it off. */
PFV_API int pfv_dec_set_output_dev(pfv_dec_session *s, uint8_t *frames_out_dev);
/* The same with `stride_bytes` (>= pfv_frame_bytes; 0 = packed) between the retframes of consecutive slots, and the slot window
 * of the *_dev calls -- the decoder-side halves of the GOP-batched use described at pfv_enc_session_set_window
 * (decode_plane_into overwrites the whole framebuffer, src/common.rs:477-496): with stride = G * pfv_frame_bytes and
 * frames_out_dev = first frame + t * pfv_frame_bytes the decoded stream appears in display order. */
PFV_API int pfv_dec_set_output_strided_dev(pfv_dec_session *s, uint8_t *frames_out_dev, size_t stride_bytes);
PFV_API int pfv_dec_session_set_window(pfv_dec_session *s, int first, int count);
/* 1 (default): packet payloads come from the device entropy stage; 0: from the host serialisers.  Same bytes. */
PFV_API int pfv_encoder_set_device_entropy(pfv_encoder *e, int on);
/* ------------------------------------------------------------------ batch encoder (n streams per step, pipelined)  [B]
 * n independent streams of one geometry encoded together -- the reference runs one Encoder per stream (src/enc.rs:12-26);
 * every writer receives exactly the bytes an Encoder of its own would have written.  Per frame step: ONE upload of all
 * frames (on a copy stream, overlapping the host-side collection of the previous step), one kernel launch per stage for
 * all streams, one download of all payloads.  Packets reach the writers one step late; finish flushes.
 *   write != NULL: called with each stream's header / packets in stream order (from the thread calling encode / finish);
 *   write == NULL: the library keeps each stream's bytes until pfv_batch_encoder_take hands them over. */
typedef struct pfv_batch_encoder pfv_batch_encoder;
typedef void (*pfv_write_cb)(void *user, int stream, const uint8_t *data, size_t len);
PFV_API int pfv_batch_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, int n_streams,
                                     pfv_write_cb write, void *user, pfv_batch_encoder **out);
/* page-locked [n_streams][pfv_frame_bytes] array to fill for the NEXT encode call (two alternate; valid for that call only) */
PFV_API uint8_t *pfv_batch_encoder_frames(pfv_batch_encoder *b);
/* one frame step for all streams.  frames == NULL: the array from pfv_batch_encoder_frames; else the caller's own
 * [n_streams][frame_bytes] buffer (ideally from pfv_host_alloc), free again when the call returns.  pframe: 0 = i-frames
 * (Encoder::encode_iframe, src/enc.rs:75-123), 1 = p-frames (:125-173).  Returns when the step is enqueued. */
PFV_API int pfv_batch_encoder_encode(pfv_batch_encoder *b, int pframe, const uint8_t *frames);
PFV_API int pfv_batch_encoder_flush(pfv_batch_encoder *b);
PFV_API int pfv_batch_encoder_finish(pfv_batch_encoder *b);
PFV_API int pfv_batch_encoder_take(pfv_batch_encoder *b, int stream, const uint8_t **data, size_t *len);
PFV_API void pfv_batch_encoder_destroy(pfv_batch_encoder *b);

/* ------------------------------------------------------------------ batch decoder (n streams per step, pipelined)  [B]
 * n `.pfv` byte streams of one geometry and one packet-type pattern decoded together: per step the packets are bit-parsed on
 * n_threads worker threads (one task per stream; 0 = on the calling thread), one kernel launch decodes all streams, one copy
 * brings the frames back; the parse of step t+1 overlaps the device work of step t.  `streams[k]` must stay valid while the
 * decoder lives (the reference's R: Read + Seek).  Frames, their order and the error codes are those of n independent
 * Decoder::advance_frame loops (src/dec.rs:169-224) run in lockstep. */
typedef struct pfv_batch_decoder pfv_batch_decoder;
PFV_API int pfv_batch_decoder_create(pfv_ctx *ctx, const uint8_t *const *streams, const size_t *lens, int n_streams, int n_threads,
                                     pfv_batch_decoder **out);
PFV_API int pfv_batch_decoder_width(const pfv_batch_decoder *b);
PFV_API int pfv_batch_decoder_height(const pfv_batch_decoder *b);
PFV_API int pfv_batch_decoder_framerate(const pfv_batch_decoder *b);
/* steps so far whose coefficient lists overflowed (more than 1 non-zero in 4) and were parsed / uploaded in the dense form */
PFV_API long pfv_batch_decoder_dense_steps(const pfv_batch_decoder *b);
/* as pfv_decoder_entropy_counts: a step whose payloads reach 64 KiB (every step under PFV_ENTROPY_DECODE_DEVICE) goes through the device's
 * entropy stage, the pool then only reads tables and block headers */
PFV_API void pfv_batch_decoder_entropy_counts(const pfv_batch_decoder *b, long counts_out[2]);
/* 1: *frames_out = [n_streams][pfv_frame_bytes] decoded frames (page-locked, valid until the call after next); 2: a step of drop
 * frames; 0: end of the streams; negative: error (PFV_ERR_FORMAT also when packet types or q-table indices diverge between streams) */
PFV_API int pfv_batch_decoder_advance(pfv_batch_decoder *b, const uint8_t **frames_out);
PFV_API void pfv_batch_decoder_destroy(pfv_batch_decoder *b);

/* ------------------------------------------------------------------ GOP-batched encoder / decoder of ONE stream  [B]
 * enc::Encoder (src/enc.rs:12-188) and dec::Decoder (src/dec.rs:15-224) with the same calls, bytes and frames as pfv_encoder /
 * pfv_decoder, but with the independent GOPs of the stream as the slots of every kernel launch: encode_iframe never reads
 * prev_frame and overwrites every plane of it (src/enc.rs:84-97), decode_plane_into overwrites the framebuffer
 * (src/common.rs:477-496), so the runs I P P ... of one stream can be worked on side by side -- frame t of every run of a batch
 * in ONE launch per stage.  A single 4K stream then fills the device the way 20 streams do.
 *   max_gops        runs ("groups") per batch = slots per launch; a group starts at every i-frame
 *   max_gop_frames  frames of a group inside one batch; a longer run continues in the next batch (its reference frame is
 *                   carried over on the device), and so does a stream that starts with p-frames
 *   payload_budget  device bytes for the packet payloads of one batch.  0: twice the batch's raw frame bytes (real content stays below
 *                   1.6 x); a batch that outgrows that all the same is encoded again frame by frame and the arena grows -- like
 *                   Encoder::encode_pframe (src/enc.rs:125-173) the object cannot fail for size.  An explicit budget is kept as given:
 *                   PFV_ERR_NOMEM from the call that completes a batch whose payloads do not fit it
 * Encoder: the planes may be reused when an encode call returns; frames are uploaded on a copy stream while the kernels of the
 * previous batch run.  A packet reaches pfv_gop_encoder_drain when its batch is complete (max_gops groups seen, flush, finish);
 * the byte stream is the one pfv_encoder writes.  After an error the stream is incomplete and every call returns PFV_ERR_STATE.
 * Decoder: one scan of the packet headers (type:u8, len:u32, src/dec.rs:179-180) cuts a batch.  The packet payloads are read
 * either on the DEVICE (PFV_OPT_ENTROPY_DECODE, the default when the batch's coefficient arrays fit: the host only reads each
 * packet's table, q indices and block headers -- n_threads workers + the caller -- and the run streams are read by the k_entd_*
 * kernels, step t + 1 on streams of their own while step t is decoded and its frames travel to the host) or by the host parser
 * pool (the packets of a frame step bit-parsed in parallel, step t + 1 under the device work of step t).  Frames are delivered in
 * stream order, with the results (1 / 0 / error) the sequential loop gives call by call -- a packet that does not parse leaves
 * the framebuffer alone, and the frames behind a failed i-frame decode against the previous run's last frame, as they do there.
 * y / u / v of the callback are one packed frame (u == y + w*h, v == u + (w/2)*(h/2)) and stay valid until the call that starts
 * the next batch. */
typedef struct pfv_gop_encoder pfv_gop_encoder;
typedef struct pfv_gop_decoder pfv_gop_decoder;
typedef struct pfv_iovec { const uint8_t *data; size_t len; } pfv_iovec;
PFV_API int pfv_gop_encoder_create(pfv_ctx *ctx, int width, int height, int framerate, int quality, int max_gops, int max_gop_frames,
                                   size_t payload_budget, pfv_gop_encoder **out);
PFV_API int pfv_gop_encoder_encode_iframe(pfv_gop_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v);
PFV_API int pfv_gop_encoder_encode_pframe(pfv_gop_encoder *e, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* the same for a packed frame (Y | U | V, pfv_frame_bytes) that already lies in DEVICE memory (frames a renderer or another kernel left in
 * HBM): nothing crosses PCIe on the way in.  Ordered on the context's stream like every *_dev call: the frame is read behind the work
 * enqueued there before the call and may be overwritten by work enqueued there after it; no host wait.  (The frame is COPIED on that stream
 * into the batch being filled; the batches' kernels run on a stream the encoder owns, so the copies of one batch run under the kernels of the
 * batch before it.  PFV_GOP_TRACE=1 in the environment: a host-side log of the encoder's submits, step downloads and arrivals on stderr.) */
PFV_API int pfv_gop_encoder_encode_iframe_dev(pfv_gop_encoder *e, const uint8_t *frame_dev);
PFV_API int pfv_gop_encoder_encode_pframe_dev(pfv_gop_encoder *e, const uint8_t *frame_dev);
/* on != 0: frames handed to the two calls above are taken BY REFERENCE -- the batch's kernels read them where they lie, no copy (16-byte
 * aligned frames; a misaligned one is copied as before).  The caller then promises that such a frame stays valid and unchanged until its batch
 * has been collected: until the packet of that frame has been handed out by pfv_gop_encoder_drain / _drain_iov, or pfv_gop_encoder_flush /
 * _finish has returned.  Same bytes.  (The frame-at-a-time contract of Encoder::encode_pframe, src/enc.rs:125 -- the frame is borrowed for
 * the call only -- is what forces the copy by default; a renderer that keeps a ring of frames does not need it.) */
PFV_API int pfv_gop_encoder_set_frames_by_reference(pfv_gop_encoder *e, int on);
PFV_API int pfv_gop_encoder_encode_dropframe(pfv_gop_encoder *e);
PFV_API int pfv_gop_encoder_flush(pfv_gop_encoder *e);
PFV_API int pfv_gop_encoder_finish(pfv_gop_encoder *e);
PFV_API int pfv_gop_encoder_drain(pfv_gop_encoder *e, const uint8_t **data, size_t *len);
PFV_API int pfv_gop_encoder_bytes(pfv_gop_encoder *e, const uint8_t **data, size_t *len);
/* the writer side without a copy: the bytes produced since the last drain as `count` segments in stream order (packet headers, and
 * payloads where the device-to-host copy put them, in page-locked memory) -- one write_all per segment, as the reference's W: Write
 * receives them (src/enc.rs:190-235).  Valid until the next call on this encoder. */
PFV_API int pfv_gop_encoder_drain_iov(pfv_gop_encoder *e, const pfv_iovec **iov, size_t *count);
PFV_API long pfv_gop_encoder_batches(const pfv_gop_encoder *e);
/* where the object's host time went, in seconds since creation (returns the number of entries written, <= n).
 * encoder: [0] waiting for plane uploads, [1] enqueueing batches, [2] waiting for a batch's kernels, [3] payloads device -> host,
 *          [4] packet assembly; counts: [5] device frames read by reference (pfv_gop_encoder_set_frames_by_reference), [6] batches that
 *          outgrew their payload arena and were encoded again
 * decoder: [0] header scan, [1] waiting for the packet parsers, [2] waiting for the device before a staging set is reused,
 *          [3] enqueueing, [4] waiting for a batch's last frames, [5] waiting for the device's entropy stage (PFV_OPT_ENTROPY_DECODE);
 *          counts: [6] packets whose payload the device read, [7] packets of such batches that were left to the host parser, of which
 *          [8] because the device's read had not settled within its rounds and [9] because it found the payload irregular;
 *          [10] host-parsed packets whose coefficient list outgrew its place in the pool and got a buffer of its own */
PFV_API int pfv_gop_encoder_stats(const pfv_gop_encoder *e, double *out, int n);
PFV_API void pfv_gop_encoder_destroy(pfv_gop_encoder *e);
PFV_API int pfv_gop_decoder_create(pfv_ctx *ctx, const uint8_t *data, size_t len, int max_gops, int max_gop_frames, int n_threads,
                                   pfv_gop_decoder **out);
PFV_API int pfv_gop_decoder_width(const pfv_gop_decoder *d);
PFV_API int pfv_gop_decoder_height(const pfv_gop_decoder *d);
PFV_API int pfv_gop_decoder_framerate(const pfv_gop_decoder *d);
PFV_API long pfv_gop_decoder_batches(const pfv_gop_decoder *d);
PFV_API int pfv_gop_decoder_stats(const pfv_gop_decoder *d, double *out, int n);
/* on != 0: decoded frames stay in device memory and the callback's y / u / v are DEVICE pointers (valid until the call that starts the
 * next batch; the context's stream is idle when the callback runs) -- for consumers on the GPU (the reference README's texture-out
 * wish, README.md:20): the download of the frames, the whole PCIe cost of decoding, is not paid.  Between batches only (PFV_ERR_STATE). */
PFV_API int pfv_gop_decoder_set_output_device(pfv_gop_decoder *d, int on);
/* Decoder::reset (src/dec.rs:148-152).  Like the reference's, it does not rewind the framebuffer; this decoder has decoded ahead of
 * the frames it delivered, so a stream whose first packet is a p-frame continues from the last DECODED frame after a reset. */
PFV_API int pfv_gop_decoder_reset(pfv_gop_decoder *d);
PFV_API int pfv_gop_decoder_advance_frame(pfv_gop_decoder *d, pfv_video_cb onvideo, void *user);
PFV_API int pfv_gop_decoder_advance_delta(pfv_gop_decoder *d, double delta, pfv_video_cb onvideo, void *user);
PFV_API void pfv_gop_decoder_destroy(pfv_gop_decoder *d);

/* packet payload serialisers alone (write_iframe_packet / write_pframe_packet bodies, src/enc.rs:237-320, 332-470);
 * return the payload size (0 on error); the payload is copied to `out` when it fits `cap` */
PFV_API size_t pfv_serialize_iframe_payload(const int16_t *coef, int total_blocks, uint8_t *out, size_t cap);
PFV_API size_t pfv_serialize_pframe_payload(const int8_t *mv, const uint8_t *has_coef, const int16_t *coef, int total_blocks,
                                            uint8_t *out, size_t cap);

/* packet payload parsers alone (the bit-reading halves of decode_iframe / decode_pframe, src/dec.rs:226-296, 328-417; host
 * only, no device): coef_out [total_blocks][256] is zero-filled first.  PFV_OK, PFV_ERR_FORMAT or PFV_ERR_IO. */
PFV_API int pfv_parse_iframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int16_t *coef_out,
                                     uint8_t qidx_out[3]);
PFV_API int pfv_parse_pframe_payload(const uint8_t *payload, size_t len, int total_blocks, int n_qtables, int8_t *mv_out,
                                     uint8_t *has_coef_out, int16_t *coef_out, uint8_t qidx_out[3]);
/* same, into the (flat index, value) list pfv_dec_*_sparse take; returns 1 when more than `cap` pairs would be needed */
PFV_API int pfv_parse_payload_sparse(int is_pframe, const uint8_t *payload, size_t len, int total_blocks, int n_qtables,
                                     int8_t *mv_out, uint8_t *has_coef_out, uint32_t *idx_out, int16_t *val_out, size_t cap,
                                     size_t *n_out, uint8_t qidx_out[3]);
/* Packets are independent bit streams: up to n_threads of them are parsed (src/dec.rs:226-296, 328-417) ahead of the
 * one being decoded, on worker threads; 0 = parse inline.  Default min(4, hardware threads - 1).  Frames, their order
 * and the error returned by each advance call are those of the sequential loop (src/dec.rs:169-224). */
PFV_API int pfv_decoder_set_lookahead(pfv_decoder *d, int n_threads);
/* The run streams of a packet are read on the DEVICE (k_entd_*, see PFV_OPT_ENTROPY_DECODE: taken from the context when the decoder is
 * created) for payloads of 64 KiB and more -- every payload under PFV_ENTROPY_DECODE_DEVICE, none under _HOST; the look-ahead threads
 * then only read tables and block headers.  counts_out[0]: packets the device read so far, [1]: packets its stage was not certain
 * about and left to the host parser.  Same frames and results either way. */
PFV_API void pfv_decoder_entropy_counts(const pfv_decoder *d, long counts_out[2]);
/* on != 0: the decoded frame stays in device memory; the callback's y / u / v are DEVICE pointers to the packed frame (u == y + w*h,
 * v == u + (w/2)*(h/2)), valid until the next advance call (see pfv_gop_decoder_set_output_device) */
PFV_API int pfv_decoder_set_output_device(pfv_decoder *d, int on);

#ifdef __cplusplus
}
#endif
#endif /* PFV_HIP_EXT_H */
