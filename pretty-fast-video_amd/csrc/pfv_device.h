// pfv_device.h -- shared definitions between the HIP kernels (pfv_kernels.hip) and the
// C-ABI host layer (pfv_capi.cpp).  gfx950 only.
#pragma once
#include <stdint.h>

// wavefronts per SIMD a kernel is compiled for (register budget); nothing on the CPU emulator build of the sources
#ifdef PFV_HIPEMU
#define PFV_WAVES_PER_EU(n)
#else
#define PFV_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

namespace pfv {

constexpr int kStripMB = 8;     // macroblocks per wavefront: a 128 x 16 pixel strip, 8 lanes per macroblock
#ifndef PFV_STRIPS_PER_WG
#define PFV_STRIPS_PER_WG 4     // tuning constant (measured: 2 and 4 within 1 % of each other)
#endif
constexpr int kStripsPerWG = PFV_STRIPS_PER_WG; // wavefronts (= strips) per workgroup
constexpr int kThreads = 64 * kStripsPerWG;

// Per-plane quantiser constants, prepared on the host from one reference q-table
// (int32_t[64], raster order, entries in [1,65535]).
//   rcp[i]  : float reciprocal of q[i], biased up by 2^-21 so that
//             trunc(float(n) * rcp[i]) == n / q[i] (Rust truncating `/`, src/dct.rs:95)
//             for every |n| <= 2^15 (proved exhaustively in tests/test_quant_recip.py).
//   deq[i]  : SCALE[z] * q[z] with z = INV_ZIGZAG[i] -- the decode side indexes its tables
//             by zigzag POSITION (src/dct.rs:78-82), reproduced on purpose.
struct QTab {
    float rcp[64];
    int32_t deq[64];
};

struct PlaneGeom {
    int w, h;        // source (unpadded) plane dims            (VideoPlane.width/height)
    int pw, ph;      // padded dims, multiples of 16            (src/common.rs:352-353)
    int bw, bh;      // macroblocks                              (src/common.rs:358-359)
    int strips_x;    // ceil(bw / kStripMB)
    int strip0;      // first strip index of this plane inside one frame
    int tiles_y;     // ceil(bh / kStripsPerWG): p-frame encode tiles are 4 vertically stacked strips
    int tile0;       // first tile index of this plane inside one frame
    int mb0;         // first macroblock index of this plane inside one frame
    int qsel;        // which QTab of the launch this plane uses
    int clear;       // pad colour: 0 luma, 128 chroma           (src/enc.rs:84-90)
    int fast_src;    // 1 when 16-byte vector loads of the source plane are legal
    long src_off;    // byte offset of the plane inside one unpadded frame
    long pad_off;    // byte offset of the plane inside one padded frame
};

struct FrameGeom {
    PlaneGeom p[3];
    int n_planes;
    int strips_per_frame;
    int tiles_per_frame;
    int mbs_per_frame;
    int n_streams;
    long src_frame_bytes;   // stride between streams, unpadded frames
    long pad_frame_bytes;   // stride between streams, padded frames
    // Encoders only, nullptr otherwise: the launch's source frames by POINTER instead of base + stream * src_frame_bytes -- slot s reads the
    // packed frame at src_slots[s] (a table in device memory).  pfv_gop_encoder's frames that already lie in device memory, taken by
    // reference (pfv_gop_encoder_set_frames_by_reference) instead of copied into the batch.
    const uint8_t *const *src_slots;
};
// the source frame of `stream` (wave-uniform: the table entry is one scalar load)
__host__ __device__ inline const uint8_t *frame_src(const FrameGeom &g, const uint8_t *src, int stream)
{
    return g.src_slots ? g.src_slots[stream] : src + (long)stream * g.src_frame_bytes;
}

// Coefficient lists (round 5): a frame's non-zero coefficients instead of its dense [macroblock][256] array -- what the decode kernels
// take from the decoders' entropy stage (k_entd_emit) and from the host parser (ListSink), so that nothing is cleared, written sparsely and
// read back in full.  The reference expands runs straight into the macroblock it is about to decode (src/dec.rs:258-296, 378-417); here a
// wavefront expands its strip's entries into its LDS zigzag stage.
//   entry      value (i16) << 16 | (macroblock index & 255) << 8 | position in the macroblock (0..255, subblock-major zigzag order);
//              ascending by (macroblock, position): the order the run streams are read in;
//   count      per macroblock m, and one more behind the last: the number of entries that belong to macroblocks BEFORE m -- an exclusive
//              prefix over the frame, so macroblock m owns entries [count[m], count[m + 1]) and a strip of neighbouring macroblocks one
//              contiguous span (a macroblock without coefficients, or one a p-frame skips, owns none).
struct CoefLists {
    const uint32_t *const *entries;    // [stream]: the stream's list
    const uint32_t *counts;            // [stream][mbs_per_frame + 1]
};
constexpr uint32_t coef_entry(uint32_t mb, uint32_t pos, int16_t value) { return ((uint32_t)(uint16_t)value << 16) | ((mb & 255u) << 8) | (pos & 255u); }

}  // namespace pfv
