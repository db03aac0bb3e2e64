"""Proof by exhaustion that the kernels' quantiser division is the reference's truncating integer division.

csrc/pfv_capi.hip:make_qtab builds  rcp = fl(fl(1/q) * (1 + 2^-21))  in f32 and the kernels compute
(int)(float(n) * rcp)  (v_cvt_f32_i32, v_mul_f32, v_cvt_i32_f32 = truncate).  The reference computes n / q on
i32 (src/dct.rs:95, truncation toward zero).  Both are evaluated here with the same IEEE f32 operations in
numpy for EVERY q in [1, 65535] and every |n| <= 8192 (the encoder's |n| is <= 5160 for u8 input), and for
|n| up to 2^15 on a q subset."""
import numpy as np


def rcp_table(q: np.ndarray) -> np.ndarray:
    r = (np.float32(1.0) / q.astype(np.float32)).astype(np.float32)
    return (r * np.float32(1.000000476837158203125)).astype(np.float32)


def check(n: np.ndarray, q: np.ndarray):
    r = rcp_table(q)
    got = (n.astype(np.float32)[None, :] * r[:, None]).astype(np.float32).astype(np.int64)   # astype(int) truncates toward zero
    want = np.sign(n)[None, :] * (np.abs(n)[None, :] // q[:, None])
    bad = np.argwhere(got != want)
    assert bad.size == 0, f"q={q[bad[0][0]]} n={n[bad[0][1]]}: got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"


def test_all_q_small_n():
    n = np.arange(-8192, 8193, dtype=np.int64)
    for lo in range(1, 65536, 2048):
        check(n, np.arange(lo, min(lo + 2048, 65536), dtype=np.int64))


def test_q_subset_full_n_range():
    n = np.arange(-32768, 32769, dtype=np.int64)
    q = np.unique(np.concatenate([np.arange(1, 600), np.array([1023, 1024, 1025, 4095, 4096, 4097, 32767, 32768, 65534, 65535]),
                                  np.random.default_rng(0).integers(1, 65536, 300)])).astype(np.int64)
    check(n, q)


def test_encoder_bound_on_n():
    """|n| = |(m * SCALE) >> 16| <= 5160 needs |m| <= 240 * 32768: the 2-D transform's gain on u8 input
    (per 1-D pass the L1 gain of any output is <= 12.5) -- checked on adversarial blocks"""
    import pfv_oracle_np as onp
    rng = np.random.default_rng(1)
    worst = 0
    blocks = [np.full((8, 8), 255), np.zeros((8, 8)), ((np.add.outer(np.arange(8), np.arange(8)) & 1) * 255)]
    blocks += [rng.choice([0, 255], (8, 8)) for _ in range(300)]
    for b in blocks:
        m = onp.fdct2d(((b.astype(np.int64) - 128) << 8)[None])[0]
        n = (m.reshape(64) * onp.DCT_SCALE_FACTOR.astype(np.int64)) >> 16
        worst = max(worst, int(np.abs(n).max()))
        assert np.abs(m).max() < 2 ** 23
    assert worst <= 5160
