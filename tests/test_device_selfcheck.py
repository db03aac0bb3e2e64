"""The f32 arithmetic the encoder kernels execute, checked ON THE DEVICE against integer arithmetic (csrc/pfv_selfcheck.h).

tests/test_quant_recip.py and tests/test_float_exact.py prove in numpy, on the host, that the float quantiser and the float
butterflies of k_enc_iframe<true> / k_enc_pframe<true> are the reference's integer arithmetic (src/dct.rs:88-99, 176-293).  Here the
device functions those kernels inline (quant_scale = v_cvt_i32_f32 + v_mul_hi_i32_i24 + v_cvt_f32_i32, quant_div = v_pk_mul_f32 +
v_trunc_f32 + the 1.5 * 2^23 add whose low 16 bits are stored, ffdct8 / fidct8 = v_pk_add / mul / fma_f32 + v_trunc_f32) run on the
MI355X itself, exhaustively where the operand space allows it:
  part 0   every n in [-8192, 8192] x every q in [1, 65535] (1.07 G divisions) vs i32 n / q and the `as i16` store
  part 1   every |m| < 2^23 x every DCT_SCALE_FACTOR vs (m * SCALE) >> 16
  part 2   the composed quantiser: every |m| < 2^23 x every SCALE x 24 spread-out q
  part 3   2^20 random 8x8 blocks (pixels / residuals / full-swing patterns) through the closed loop of both forms at every
           quality's tables: 2^20 x (16 + 22 x 16) 1-D transforms, every intermediate compared
  part 4   the L1 worst-case blocks behind enc_float_exact (forward and inverse), every quality
  part 5   trunc(delta / 2) << 8 by two fused multiply-adds (residual_f), every (source, prediction) byte pair
  part 6   the i-frame pixel without v_floor (v_cvt_pk_u8_f32 rounds to nearest even and saturates), every |x| < 2^24
The emulator runs of the same entry point (small ranges) keep the check itself honest in the GPU-less container."""
import ctypes

import pytest


# values compared per 8x8 block in parts 3 / 4a: 11 qualities x (64 forward outputs + 2 tables x (64 quotients + 64 inverse outputs + 64 `>> 8`))
PER_BLOCK = 11 * (64 + 2 * (64 + 128))


def run_part(ctx, part, arg=0):
    lib = ctx._lib
    fn = lib.pfv_selfcheck_float_path            # exported by libpfv_hip.so, declared in csrc/pfv_selfcheck.h (not in pfv_hip.h)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                   ctypes.POINTER(ctypes.c_int64)]
    checked, bad = ctypes.c_uint64(), ctypes.c_uint64()
    first = (ctypes.c_int64 * 4)()
    ctx.check(fn(ctx.handle, part, arg, ctypes.byref(checked), ctypes.byref(bad), first))
    assert bad.value == 0, f"part {part}: {bad.value} of {checked.value} evaluations differ; first: operands {first[0]}, {first[1]:#x}: got {first[2]}, want {first[3]}"
    return checked.value


@pytest.mark.gpu
def test_gpu_quantiser_division_exhaustive(gpu_ctx):
    assert run_part(gpu_ctx, 0) == 65535 * 2 * 8193           # 1 073 856 510 (n, q) pairs


@pytest.mark.gpu
def test_gpu_quantiser_scaling_exhaustive(gpu_ctx):
    assert run_part(gpu_ctx, 1) == (1 << 23) * 2 * 10


@pytest.mark.gpu
def test_gpu_quantiser_composed(gpu_ctx):
    assert run_part(gpu_ctx, 2) == (1 << 23) * 2 * 10 * 24


@pytest.mark.gpu
def test_gpu_float_butterflies_random_blocks(gpu_ctx):
    n = run_part(gpu_ctx, 3, 1 << 20)
    assert n == (1 << 20) * PER_BLOCK


@pytest.mark.gpu
def test_gpu_float_butterflies_worst_case_blocks(gpu_ctx):
    assert run_part(gpu_ctx, 4) == 2 * 256 * PER_BLOCK + 128 * 44 * 2 * 128


@pytest.mark.gpu
def test_gpu_residual_by_fma_exhaustive(gpu_ctx):
    assert run_part(gpu_ctx, 5) == 2 * 256 * 256


@pytest.mark.gpu
def test_gpu_iframe_pixel_without_floor_exhaustive(gpu_ctx):
    assert run_part(gpu_ctx, 6) == (1 << 24) * 2


def test_emu_selfcheck_small_ranges(emu_ctx):
    """the same entry point on the CPU emulator (IEEE f32 on the host): small slices of every part"""
    assert run_part(emu_ctx, 0, 96) == 96 * 2 * 8193
    assert run_part(emu_ctx, 1, 64) == 64 * 256 * 2 * 10
    assert run_part(emu_ctx, 2, 8) == 8 * 256 * 2 * 10 * 24
    assert run_part(emu_ctx, 3, 256) == 256 * PER_BLOCK
    assert run_part(emu_ctx, 4) == 2 * 256 * PER_BLOCK + 128 * 44 * 2 * 128
    assert run_part(emu_ctx, 5) == 2 * 256 * 256
    assert run_part(emu_ctx, 6, 1024) == 1024 * 256 * 2
