"""Why the encoders may run their transforms in f32 (csrc/pfv_kernels.hip, "the same transforms in f32"): every value is an integer
below 2^24, so float arithmetic reproduces the integer butterflies bit for bit.  This file proves the two bounds the kernels rely on
with the numpy oracle -- the same L1 argument pfv_capi.hip's enc_float_exact() evaluates for the tables a session is created with --
and the -m gpu / emulator tests below drive sessions with the most hostile 8-bit content there is against the (integer) oracle."""
import os

import numpy as np
import pytest

import pfv_oracle_np as onp
import parity_cases as pc


def _abs_matrix(fn):
    M = np.zeros((8, 8))
    for k in range(8):
        e = np.zeros(8, dtype=np.int64)
        e[k] = 1 << 20                      # large enough that no division of the 1-D transform truncates
        M[:, k] = np.abs(fn(e[None])[0]) / float(1 << 20)
    return M


F, I = _abs_matrix(onp.fdct), _abs_matrix(onp.idct)


def test_forward_transform_stays_below_2_pow_24_for_any_8bit_input():
    """|fdct2d| <= amplitude * (row L1 norm)^2; i-frames feed (px - 128) << 8, p-frames trunc(delta / 2) << 8 (src/common.rs:291, :304)"""
    R = F.sum(1)
    worst = 128 * 256 * np.outer(R, R).max()
    assert worst < 2 ** 22                  # 2.5 M: two bits of headroom
    # every partial sum inside the butterflies is bounded by the same L1 argument applied to fewer terms


@pytest.mark.parametrize("quality", range(11))
def test_closed_loop_inverse_stays_below_2_pow_23_for_quality_tables(quality):
    """largest coefficient the encoder can produce at each position, dequantised the way decode indexes its tables
    (src/dct.rs:78-82), pushed through |idct| columns then rows with all 64 maxima at once (+ slack for the truncations)"""
    il, ic, pl, pcq, _ = onp.qtables(quality)
    S = onp.DCT_SCALE_FACTOR.astype(np.float64)
    z = onp.INV_ZIGZAG_TABLE
    R = F.sum(1)
    for amp, q in ((128 * 256, il), (128 * 256, ic), (127 * 256, pl), (127 * 256, pcq)):
        M = amp * np.outer(R, R).reshape(64)
        c = np.floor(np.floor(M * S / 65536) / q)
        D = (c * S[z] * q[z]).reshape(8, 8)
        col = I @ D + 16
        out = col @ I.T + 16
        assert max(M.max(), D.max(), col.max(), out.max()) < 2 ** 23


def _hostile_frames(w, h, n, seed):
    """8-bit content that maximises transform magnitudes: full-swing checkerboards at several periods, hard edges, 0/255 noise,
    all-black to all-white steps between frames"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    pats = [((xx // p + yy // p) & 1) * 255 for p in (1, 2, 4, 8)]
    pats += [(xx & 1) * 255, (yy & 1) * 255, rng.integers(0, 2, (h, w)) * 255, np.full((h, w), 255), np.zeros((h, w), int),
             ((xx * 7 + yy * 13) % 256)]
    out = []
    for t in range(n):
        y = pats[(seed + t) % len(pats)].astype(np.uint8)
        u = pats[(seed + 3 * t + 1) % len(pats)][: h // 2, : w // 2].astype(np.uint8)
        v = pats[(seed + 5 * t + 2) % len(pats)][: h // 2, : w // 2].astype(np.uint8)
        out.append(np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]))
    return out


def _check_hostile_sessions(pkg, ctx, oracle, qualities, w=80, h=48, n_frames=5):
    for quality in qualities:
        frames = _hostile_frames(w, h, n_frames, seed=quality)
        enc = pkg.EncoderSession(ctx, w, h, quality, 1)
        oenc = oracle.encoder(w, h, quality)
        for t, f in enumerate(frames):
            if t % 3 == 0:
                coef = enc.encode_iframe(f[None])
                assert np.array_equal(coef[0], oenc.encode_iframe(f)), (quality, t)
            else:
                mv, has, coef = enc.encode_pframe(f[None])
                omv, ohas, ocoef = oenc.encode_pframe(f)
                assert np.array_equal(mv[0], omv) and np.array_equal(has[0], ohas) and np.array_equal(coef[0], ocoef), (quality, t)
            assert np.array_equal(enc.prev_frame()[0], oenc.prev_frame()), (quality, t)
        enc.close()


def test_emu_hostile_content_sessions(pkg, emu_ctx, oracle):
    _check_hostile_sessions(pkg, emu_ctx, oracle, (0, 1, 5, 10))


def test_emu_integer_transform_fallback(pkg, emu_ctx, oracle):
    """pfv_ctx_set_option(PFV_OPT_ENC_TRANSFORM, PFV_ENC_TRANSFORM_INT) (or a table that fails the bound) keeps the integer
    encoder kernels, in sessions and in the plane-level operators: same bytes"""
    L = pkg._lib
    assert emu_ctx.get_option(L.PFV_OPT_ENC_TRANSFORM) == L.PFV_ENC_TRANSFORM_AUTO
    emu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_INT)
    try:
        assert emu_ctx.get_option(L.PFV_OPT_ENC_TRANSFORM) == L.PFV_ENC_TRANSFORM_INT
        _check_hostile_sessions(pkg, emu_ctx, oracle, (0, 7))
        pc.check_session(pkg, emu_ctx, oracle, 64, 48, 5, n_streams=2, n_frames=3)
        pc.check_golden(pkg, emu_ctx, oracle)
    finally:
        emu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_AUTO)
    with pytest.raises(pkg.PfvError):
        emu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, 7)
    with pytest.raises(pkg.PfvError):
        emu_ctx.set_option(99, 0)


@pytest.mark.gpu
def test_gpu_hostile_content_sessions(pkg, gpu_ctx, oracle):
    _check_hostile_sessions(pkg, gpu_ctx, oracle, range(11), w=208, h=112, n_frames=7)


@pytest.mark.gpu
def test_gpu_integer_transform_fallback(pkg, gpu_ctx, oracle):
    L = pkg._lib
    gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_INT)
    try:
        _check_hostile_sessions(pkg, gpu_ctx, oracle, (0, 5, 10), w=208, h=112, n_frames=4)
        pc.check_session(pkg, gpu_ctx, oracle, 320, 240, 5, n_streams=2, n_frames=4)
        pc.check_golden(pkg, gpu_ctx, oracle)
        pc.check_trap_vectors(pkg, gpu_ctx, oracle)
    finally:
        gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_AUTO)
