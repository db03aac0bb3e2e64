"""CPU suite: the C oracle against (a) the committed known-answer vectors (tests/golden), (b) the
independently written numpy restatement, (c) structural properties.  Bit-exact."""
import os

import numpy as np
import pytest

import pfv_oracle_np as onp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_vectors.npz"))


def test_reference_test_inputs_known_answers(oracle):
    """inputs of the reference's own (print-only) tests: src/lib.rs:38 ramp, src/lib.rs:61-66 block"""
    assert oracle.fdct8(GOLD["ramp"]).tolist() == [71680, -40000, 0, -6400, 0, -1280, 0, -320]
    assert np.array_equal(oracle.fdct8(GOLD["ramp"]), GOLD["ramp_fdct"])
    assert np.array_equal(oracle.idct8(GOLD["ramp_fdct"]), GOLD["ramp_idct"])
    q = GOLD["lib_q"]
    assert np.array_equal(q, oracle.qtables(5)[0])            # the test's q-table IS the quality-5 luma intra table
    out = oracle.encode_subblock(GOLD["lib_block"], q)
    assert np.array_equal(out, GOLD["lib_quant"])
    assert out[:12].tolist() == [-150, 2, 5, 0, 0, -2, 0, 0, 0, 0, 0, -1]
    assert int(GOLD["lib_prequant"][0]) == -1538048
    rec = oracle.decode_subblock(out, q)
    assert np.array_equal(rec, GOLD["lib_recon"])
    assert ((rec.astype(int) - GOLD["lib_block"]) ** 2).mean() < 25.0


@pytest.mark.parametrize("quality", [0, 2, 5, 10])
@pytest.mark.parametrize("table", ["intra_l", "intra_c", "inter_l"])
def test_golden_subblocks(oracle, quality, table):
    il, ic, pl, pc, _ = oracle.qtables(quality)
    q = {"intra_l": il, "intra_c": ic, "inter_l": pl}[table]
    enc = np.stack([oracle.encode_subblock(b, q) for b in GOLD["sub_px"]])
    assert np.array_equal(enc, GOLD[f"q{quality}_{table}_enc"])
    encd = np.stack([oracle.encode_subblock_delta(d, q) for d in GOLD["sub_delta"]])
    assert np.array_equal(encd, GOLD[f"q{quality}_{table}_encdelta"])
    dec = np.stack([oracle.decode_subblock(c, q) for c in enc])
    assert np.array_equal(dec, GOLD[f"q{quality}_{table}_dec"])


def test_golden_pframe_case(oracle):
    """64x48 two-frame case: translation (+5,-3) found for interior blocks, border blocks differ"""
    il, _, pl, _, px_err = oracle.qtables(5)
    c0, bw, bh = oracle.encode_plane(GOLD["pf_f0"], il, 0)
    assert np.array_equal(c0, GOLD["pf_c0"])
    rec0 = oracle.decode_plane(c0, bw, bh, il)
    assert np.array_equal(rec0, GOLD["pf_rec0"])
    mv, has, c1 = oracle.encode_plane_delta(GOLD["pf_f1"], rec0, pl, px_err, 0)
    assert np.array_equal(mv, GOLD["pf_mv"]) and np.array_equal(has, GOLD["pf_has"]) and np.array_equal(c1, GOLD["pf_c1"])
    assert [5, -3] in mv.tolist()
    rec1 = oracle.decode_plane_delta(mv, has, c1, bw, bh, pl, rec0)
    assert np.array_equal(rec1, GOLD["pf_rec1"])
    ext = np.stack([oracle.encode_subblock_delta(GOLD["ext_delta"], q) for q in (oracle.qtables(1)[2], oracle.qtables(10)[3])])
    assert np.array_equal(ext, GOLD["ext_out"])


def test_tables_and_qtable_derivation(oracle):
    assert np.array_equal(onp.ZIGZAG_TABLE[onp.INV_ZIGZAG_TABLE], np.arange(64))     # mutually inverse (dct.rs:39-47)
    for quality in range(11):
        c = oracle.qtables(quality)
        n = onp.qtables(quality)
        for a, b in zip(c[:4], n[:4]):
            assert np.array_equal(a, b)
        assert c[4] == n[4] == quality * 1.5
        # the f32 derivation (enc.rs:48-51) equals the integer forms noted in SURVEY 8a-20
        assert np.array_equal(c[0], np.maximum(1, (onp.Q_TABLE_INTRA * quality) >> 3))
        assert np.array_equal(c[1], np.maximum(1, (onp.Q_TABLE_INTRA * quality) >> 2))
        assert np.all(c[2] == max(1, 2 * quality)) and np.all(c[3] == max(1, 4 * quality))


@pytest.mark.parametrize("w,h", [(64, 48), (50, 38), (16, 16), (33, 17)])
def test_c_oracle_equals_numpy_oracle(oracle, w, h):
    rng = np.random.default_rng(w * 100 + h)
    px = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for quality in (0, 3, 5, 10):
        il, ic, pl, pc, px_err = oracle.qtables(quality)
        c_c, bw, bh = oracle.encode_plane(px, il, 9)
        c_n, bw2, bh2 = onp.encode_plane(px, il, 9)
        assert (bw, bh) == (bw2, bh2) and np.array_equal(c_c, c_n)
        d = oracle.decode_plane(c_c, bw, bh, il)
        assert np.array_equal(d, onp.decode_plane(c_n, bw, bh, il))
        ref = np.roll(np.clip(d.astype(int) + rng.integers(-4, 5, d.shape), 0, 255).astype(np.uint8), (1, -2), (0, 1))
        ref = np.ascontiguousarray(ref)
        mv, has, cf = oracle.encode_plane_delta(px, ref, pc, px_err, 9)
        mv2, has2, cf2 = onp.encode_plane_delta(px, ref, pc, px_err, 9)
        assert np.array_equal(mv, mv2) and np.array_equal(has, has2) and np.array_equal(cf, cf2)
        assert np.array_equal(oracle.decode_plane_delta(mv, has, cf, bw, bh, pc, ref), onp.decode_plane_delta(mv2, has2, cf2, bw, bh, pc, ref))


def test_threaded_oracle_equals_serial(oracle):
    """the fork/join stand-in for rayon changes nothing (collect() keeps index order, common.rs:374-378)"""
    rng = np.random.default_rng(5)
    px = rng.integers(0, 256, (80, 112), dtype=np.uint8)
    il, _, pl, _, px_err = oracle.qtables(4)
    a, bw, bh = oracle.encode_plane(px, il, 0, threads=1)
    b, _, _ = oracle.encode_plane(px, il, 0, threads=5)
    assert np.array_equal(a, b)
    ref = oracle.decode_plane(a, bw, bh, il, threads=3)
    r1 = oracle.encode_plane_delta(px[::-1].copy(), ref, pl, px_err, 0, threads=1)
    r2 = oracle.encode_plane_delta(px[::-1].copy(), ref, pl, px_err, 0, threads=4)
    assert all(np.array_equal(x, y) for x, y in zip(r1, r2))


def test_properties(oracle):
    # DC gain 8 per 1-D pass: constant input -> only c0
    assert oracle.fdct8(np.full(8, 256, np.int32)).tolist() == [2048, 0, 0, 0, 0, 0, 0, 0]
    # odd symmetry of the truncating arithmetic: fdct(-x) == -fdct(x), idct likewise
    rng = np.random.default_rng(3)
    for _ in range(50):
        v = rng.integers(-40000, 40000, 8).astype(np.int32)
        assert np.array_equal(oracle.fdct8(-v), -oracle.fdct8(v))
        assert np.array_equal(oracle.idct8(-v), -oracle.idct8(v))
    # i32 wrap-around instead of UB / panic on hostile decode input
    big = np.full(8, 2**31 - 1, np.int64).astype(np.int32)
    assert np.array_equal(oracle.idct8(big), onp.idct(big[None].astype(np.int64))[0].astype(np.int32))
    # quality 0 -> every q == 1, near-lossless round trip on a smooth block
    q0 = oracle.qtables(0)[0]
    assert np.all(q0 == 1)
    blk = (np.add.outer(np.arange(8), np.arange(8)) * 6 + 40).astype(np.uint8)
    rec = oracle.decode_subblock(oracle.encode_subblock(blk, q0), q0)
    assert np.abs(rec.astype(int) - blk.reshape(-1)).max() <= 6


def test_session_oracle_encoder_decoder_agree(oracle):
    """Encoder's closed-loop prev_frame == what a Decoder reconstructs from the same coefficients"""
    from oracle_bind import OracleDecoder
    import __graft_entry__ as g
    pkg = g.load_package()
    st = pkg.SyntheticStream(48, 32)
    enc = oracle.encoder(48, 32, 5)
    dec = OracleDecoder(oracle, 48, 32, np.stack(oracle.qtables(5)[:4]))
    for t in range(4):
        if t == 0:
            dec.decode_iframe(enc.encode_iframe(st.frame(t)))
        else:
            dec.decode_pframe(*enc.encode_pframe(st.frame(t)))
        assert np.array_equal(enc.prev_frame(), dec.framebuffer())


def test_forward_dct_is_exact(oracle):
    """The kernels compute the forward DCT with plain arithmetic shifts instead of truncating divisions
    (csrc/pfv_kernels.hip::fdct8): with 8 zero fraction bits on the input, every division in the row pass acts on
    a multiple of 256 and every division in the column pass on a multiple of 16, so nothing is ever truncated.
    Checked here by evaluating the reference form (truncating, numpy oracle) and a floor-shift form on u8 blocks
    and on +-255 residual blocks, incl. all-extreme inputs."""
    def fdct_floor(v):
        i = [v[..., k].astype(np.int64) for k in range(8)]
        a0, a1, a2, a3 = i[0] + i[7], i[1] + i[6], i[2] + i[5], i[3] + i[4]
        a4, a5, a6, a7 = i[0] - i[7], i[1] - i[6], i[2] - i[5], i[3] - i[4]
        b0, b1, b2, b3 = a0 + a3, a1 + a2, a0 - a3, a1 - a2
        c0, c1 = b0 + b1, b0 - b1
        c2 = b2 + (b2 >> 2) + (b3 >> 1)
        c3 = (b2 >> 1) - b3 - (b3 >> 2)
        b4 = (a7 >> 2) + a4 + (a4 >> 2) - (a4 >> 4)
        b7 = (a4 >> 2) - a7 - (a7 >> 2) + (a7 >> 4)
        b5 = a5 + a6 - (a6 >> 2) - (a6 >> 4)
        b6 = a6 - a5 + (a5 >> 2) + (a5 >> 4)
        c4, c5, c6, c7 = b4 + b5, b4 - b5, b6 + b7, b6 - b7
        return np.stack([c0, c4, c2, c5 - c7, c1, c5 + c7, c3, c6], axis=-1)

    rng = np.random.default_rng(12)
    px = [rng.integers(0, 256, (400, 8, 8)), rng.choice([0, 255], (400, 8, 8)), np.zeros((1, 8, 8), int), np.full((1, 8, 8), 255)]
    blocks = [((b.astype(np.int64) - 128) << 8) for b in px]
    d = np.concatenate([rng.integers(-255, 256, (400, 8, 8)), rng.choice([-255, 255], (200, 8, 8))])
    blocks.append((np.sign(d) * (np.abs(d) // 2)).astype(np.int64) << 8)          # (delta / 2) << 8, common.rs:304
    for m in blocks:
        rows_ref, rows_fl = onp.fdct(m), fdct_floor(m)
        assert np.array_equal(rows_ref, rows_fl) and not (rows_ref & 15).any()     # row outputs: multiples of 16
        cols_ref = onp.fdct(np.swapaxes(rows_ref, -1, -2))
        assert np.array_equal(cols_ref, fdct_floor(np.swapaxes(rows_fl, -1, -2)))


def test_config1_plumbing_161_frames_1080p(oracle, pkg):
    """BASELINE config #1 (the reference's own CPU-runnable case; its test_frames/ fixtures are Git-LFS stubs, so the
    161 frames are synthetic 1080p): encode_iframe every 15th frame, encode_pframe otherwise, q=5, on the
    CPU oracle -> .pfv bytes -> decode; every decoded frame equals the encoder's closed-loop reconstruction.  As BASELINE.json words it:
    num_threads = 1 (src/enc.rs:54) and 161 DISTINCT frames."""
    from oracle_bind import OracleEncoder, OracleStreamDecoder, OracleStreamEncoder
    W, H, Q, N = 1920, 1080, 5, 161
    th = 1
    st = pkg.SyntheticStream(W, H)
    senc = OracleStreamEncoder(oracle, W, H, 30, Q, threads=th)
    henc = OracleEncoder(oracle, W, H, Q, threads=th)          # same hot path; exposes prev_frame
    pw = henc.prev_frame().size
    recon_sums = []
    ny, nc = W * H, (W // 2) * (H // 2)
    pwy, phy = 1920, 1088
    for t in range(N):
        f = st.frame(t)                                         # every frame its own (the texture's translation has period 23 x 17 frames)
        if t % 15 == 0:
            senc.encode_iframe(f); henc.encode_iframe(f)
        else:
            senc.encode_pframe(f); henc.encode_pframe(f)
        p = henc.prev_frame()
        y = p[:pwy * phy].reshape(phy, pwy)[:H, :W]
        u = p[pwy * phy:pwy * phy + 960 * 544].reshape(544, 960)[:H // 2, :W // 2]
        v = p[pwy * phy + 960 * 544:].reshape(544, 960)[:H // 2, :W // 2]
        recon_sums.append((int(y.sum()), int(u.sum()), int(v.sum()), bytes(y[H // 2, :64])))
    senc.finish()
    data = senc.bytes()
    assert 161 * 20 < len(data) < 161 * W * H                  # a real stream, smaller than raw
    dec = OracleStreamDecoder(oracle, data, threads=th)
    assert (dec.width, dec.height, dec.framerate) == (W, H, 30)
    n = 0
    while True:
        rc, fr = dec.advance_frame()
        assert rc >= 0
        if fr is not None:
            y, u, v = fr[:ny].reshape(H, W), fr[ny:ny + nc], fr[ny + nc:]
            assert (int(y.sum()), int(u.sum()), int(v.sum()), bytes(y[H // 2, :64])) == recon_sums[n], f"frame {n}"
            n += 1
        if rc == 0:
            break
    assert n == N
    assert pw == pwy * phy + 2 * 960 * 544


def test_colour_helpers_c_equals_numpy(oracle):
    """src/lib.rs:337-394 restated twice (C with forced f32 rounding, numpy float32): same bytes, incl. saturation"""
    import ctypes
    import parity_cases as pc
    L = pc._oracle_colour(oracle)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (512, 1024, 3)).astype(np.uint8)
    rgb[:2, :4] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 0, 255]]
    h, w = rgb.shape[:2]
    out = np.empty(w * h * 3 // 2, np.uint8)
    L.pfvo_rgb_to_yuv420(P(rgb), w, h, P(out))
    assert np.array_equal(out, onp.rgb_to_yuv420(rgb))
    assert out[0] == 0 and out[1] == 255                      # black, white
    frame = rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8)
    back = np.empty((h, w, 3), np.uint8)
    L.pfvo_yuv420_to_rgb(P(frame), w, h, P(back))
    assert np.array_equal(back, onp.yuv420_to_rgb(frame, w, h))
    assert back.min() == 0 and back.max() == 255              # the saturating casts are exercised
