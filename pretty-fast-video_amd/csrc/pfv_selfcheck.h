/* pfv_selfcheck.h -- device self-check of the f32 arithmetic the encoder kernels execute.  NOT part of the drop-in boundary
 * (include/pfv_hip.h): a maintenance entry point of libpfv_hip.so used by tests/test_device_selfcheck.py.
 *
 * k_enc_iframe<true> / k_enc_pframe<true> evaluate the reference's i32 transforms and quantiser (src/dct.rs:88-99, 176-293) in
 * f32 where that is provably the same arithmetic.  The proofs (tests/test_quant_recip.py, tests/test_float_exact.py) run in
 * numpy on the host; this entry point runs the SAME device functions the kernels inline (quant_scale, quant_div, quant_low16,
 * ffdct8, fidct8 of csrc/pfv_kernels.hip) on the GPU itself against integer arithmetic evaluated next to them:
 *   part 0  quant_div + quant_low16 for every n in [-8192, 8192] x every q in [1, 65535]        vs  n / q (i32, truncating)
 *   part 1  quant_scale for every |m| < 2^23 x every distinct DCT_SCALE_FACTOR                   vs  (m * SCALE) >> 16 (i64)
 *   part 2  the composed quantiser for every |m| < 2^23 x every SCALE x `arg` spread-out q       vs  ((m * SCALE) >> 16) / q
 *   part 3  `arg` random 8x8 blocks (pixels, residuals, full-swing patterns) through the whole closed loop of both encoder
 *           forms -- rows, columns, quantise, dequantise, inverse columns, inverse rows, >> 8 -- at every quality's four tables,
 *           every intermediate compared (the 1-D transforms run 32 x `arg` times)
 *   part 4  the L1 worst-case blocks behind enc_float_exact: sign patterns that maximise each transform output, at full
 *           amplitude, forward and inverse, every quality
 *   part 5  residual_f (trunc(delta / 2) << 8 by two fused multiply-adds) for every (source, prediction) byte pair
 *   part 6  the i-frame pixel without a floor (iframe_pixel_f + v_cvt_pk_u8_f32's round-to-nearest-even and saturation) for every
 *           inverse-transform output |x| < 2^24                                                 vs  clamp((x >> 8) + 128, 0, 255)
 * checked = evaluations performed, mismatches = how many differed (saturating at 2^32 - 1); first_bad = {operand a, operand b,
 * got, want} of the lowest-numbered failing evaluation, when there is one. */
#ifndef PFV_SELFCHECK_H
#define PFV_SELFCHECK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
struct pfv_ctx;
__attribute__((visibility("default"))) int pfv_selfcheck_float_path(struct pfv_ctx *ctx, int part, uint64_t arg, uint64_t *checked,
                                                                    uint64_t *mismatches, int64_t first_bad[4]);
#ifdef __cplusplus
}
#endif
#endif
