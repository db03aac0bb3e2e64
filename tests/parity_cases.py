"""Parity checks shared by the GPU tests (tests/test_gpu_parity.py, real MI355X through the
C ABI) and the emulator tests (tests/test_emulated_kernels.py, same kernel sources compiled
for the CPU).  Every check is bit-exact: integer / byte / index work."""
from __future__ import annotations

import numpy as np

from oracle_bind import pad16


def smooth_plane(h, w, seed):
    """deterministic smooth-ish u8 plane with fine noise (the 4-step search needs gradients)"""
    rng = np.random.default_rng(seed)
    gh, gw = h // 8 + 2, w // 8 + 2
    g = rng.integers(0, 256, (gh, gw)).astype(np.int32)
    y, x = np.arange(h), np.arange(w)
    gy, fy = (y >> 3)[:, None], (y & 7)[:, None]
    gx, fx = (x >> 3)[None, :], (x & 7)[None, :]
    t = ((8 - fy) * ((8 - fx) * g[gy, gx] + fx * g[gy, gx + 1]) + fy * ((8 - fx) * g[gy + 1, gx] + fx * g[gy + 1, gx + 1])) >> 6
    return np.clip(t + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)


def shifted_ref(px, dx, dy, seed, noise=2, clear=0):
    """a padded 'previous frame' = px shifted by (dx,dy) with a little noise on half the blocks"""
    h, w = px.shape
    ph, pw = pad16(h), pad16(w)
    rng = np.random.default_rng(seed)
    img = np.full((ph, pw), clear, dtype=np.int32)
    img[:h, :w] = px
    img = np.roll(img, (dy, dx), (0, 1))
    mask = np.repeat(np.repeat(rng.integers(0, 2, (ph // 16, pw // 16)), 16, 0), 16, 1).astype(bool)
    img = np.where(mask, img + rng.integers(-noise, noise + 1, img.shape), img)
    return np.clip(img, 0, 255).astype(np.uint8)


def check_encode_plane(pkg, ctx, oracle, px, q, clear):
    h, w = px.shape
    plane = pkg.VideoPlane.from_slice(w, h, px)
    enc = plane.encode_plane(q, clear, ctx)
    ocoef, bw, bh = oracle.encode_plane(px, q, clear)
    assert (enc.blocks_wide, enc.blocks_high) == (bw, bh)
    assert enc.width == pad16(w) and enc.height == pad16(h)
    assert np.array_equal(enc.blocks, ocoef), "encode_plane coefficients differ"
    dec = pkg.VideoPlane.decode_plane(enc, q, ctx)
    odec = oracle.decode_plane(ocoef, bw, bh, q)
    assert np.array_equal(dec.image(), odec), "decode_plane pixels differ"
    return enc, dec


def check_encode_plane_delta(pkg, ctx, oracle, px, ref, q, px_err, clear):
    h, w = px.shape
    plane = pkg.VideoPlane.from_slice(w, h, px)
    refplane = pkg.VideoPlane.from_slice(ref.shape[1], ref.shape[0], ref)
    enc = plane.encode_plane_delta(refplane, q, px_err, clear, ctx)
    omv, ohas, ocoef = oracle.encode_plane_delta(px, ref, q, px_err, clear)
    assert np.array_equal(enc.motion, omv), "motion vectors differ"
    assert np.array_equal(enc.has_coeff, ohas), "skip flags differ"
    assert np.array_equal(enc.blocks, ocoef), "residual coefficients differ"
    dec = pkg.VideoPlane.decode_plane_delta(enc, refplane, q, ctx)
    odec = oracle.decode_plane_delta(omv, ohas, ocoef, enc.blocks_wide, enc.blocks_high, q, ref)
    assert np.array_equal(dec.image(), odec), "decode_plane_delta pixels differ"
    # _into form: read-all-then-write-all into the same plane (src/common.rs:498-521)
    target = pkg.VideoPlane.from_slice(ref.shape[1], ref.shape[0], ref)
    pkg.VideoPlane.decode_plane_delta_into(enc, target, q, ctx)
    assert np.array_equal(target.image(), odec), "decode_plane_delta_into pixels differ"
    return enc, dec


def check_sparse_coded_tiles(pkg, ctx, oracle, seed=21, sizes=((256, 128), (400, 200), (130, 70), (1040, 64))):
    """p-frame planes in which only SOME macroblocks are coded, in every arrangement the tile-level compaction of k_enc_pframe
    distinguishes (src/common.rs:221-222: a skipped macroblock is not transformed; the kernel moves a tile's coded macroblocks
    together when they fit fewer wavefronts than the strips that hold them): one coded macroblock per strip, 8 / 9 / 16 / 17 per
    tile, all in one strip, random densities, ragged tiles (partial strips, missing strips) -- every output against the oracle"""
    _, _, pl, pcq, px_err = oracle.qtables(5)
    rng = np.random.default_rng(seed)
    n_cases = 0
    for (w, h) in sizes:
        base = smooth_plane(h, w, seed + w)
        ph, pw = pad16(h), pad16(w)
        bw, bh = pw // 16, ph // 16
        ref = np.zeros((ph, pw), np.uint8)
        ref[:h, :w] = base                      # prediction == source wherever the source is left alone: SSD 0 -> skipped
        patterns = []
        for dens in (0.03, 0.1, 0.25, 0.5, 0.8):
            patterns.append(rng.random((bh, bw)) < dens)
        one_per_strip = np.zeros((bh, bw), bool); one_per_strip[:, ::8] = True
        patterns.append(one_per_strip)
        col = np.zeros((bh, bw), bool); col[:, min(3, bw - 1)] = True
        patterns.append(col)
        rowp = np.zeros((bh, bw), bool); rowp[0::4, :] = True               # one full strip per tile
        patterns.append(rowp)
        for n in (8, 9, 16, 17):                                            # exactly n coded macroblocks in the first tile, scattered
            pat = np.zeros((bh, bw), bool)
            cells = [(y, x) for y in range(min(4, bh)) for x in range(min(8, bw))]
            for k in rng.permutation(len(cells))[:min(n, len(cells))]:
                pat[cells[k]] = True
            patterns.append(pat)
        for pat in patterns:
            px = base.astype(np.int32).copy()
            mask = np.repeat(np.repeat(pat, 16, 0), 16, 1)[:h, :w]
            px = np.where(mask, px + rng.integers(-60, 61, px.shape), px)
            px = np.clip(px, 0, 255).astype(np.uint8)
            for q, clear in ((pl, 0), (pcq, 128)):
                enc, _ = check_encode_plane_delta(pkg, ctx, oracle, px, ref, q, px_err, clear)
                n_cases += 1
            got = enc.has_coeff.reshape(bh, bw).astype(bool)
            assert got.sum() >= 0.5 * (pat & (np.add.outer(np.arange(bh) * 16, np.zeros(bw, int)) < h)).sum()   # the perturbed macroblocks are coded
    return n_cases


def check_session(pkg, ctx, oracle, width, height, quality, n_streams, n_frames, gop=15, threads=1, kind="pan"):
    """encode n_frames of n_streams synthetic streams (i-frame every `gop`), decode them again,
    compare everything with the oracle stream by stream."""
    streams = [pkg.SyntheticStream(width, height, seed=pkg.synth.SEED + 17 * s, kind=kind) for s in range(n_streams)]
    enc = pkg.EncoderSession(ctx, width, height, quality, n_streams)
    tabs = pkg.qtables_from_quality(quality)
    dec = pkg.DecoderSession(ctx, width, height, np.stack(tabs[:4]), n_streams)
    oencs = [oracle.encoder(width, height, quality, threads) for _ in range(n_streams)]
    stats = {"coded": 0, "mbs": 0}
    # fused retframe output of the decode kernels (pfv_dec_set_output_dev), checked against the same crop
    fused_dev = ctx.alloc(n_streams * dec.frame_bytes)
    dec.set_output_dev(fused_dev)
    fused = np.empty((n_streams, dec.frame_bytes), dtype=np.uint8)
    for t in range(n_frames):
        frames = np.stack([s.frame(t) for s in streams])
        if t % gop == 0:
            coef = enc.encode_iframe(frames)
            dec.decode_iframe(coef)
            for s in range(n_streams):
                ocoef = oencs[s].encode_iframe(frames[s])
                assert np.array_equal(coef[s], ocoef), f"frame {t} stream {s}: i-frame coefficients differ"
        else:
            mv, has, coef = enc.encode_pframe(frames)
            dec.decode_pframe(mv, has, coef)
            for s in range(n_streams):
                omv, ohas, ocoef = oencs[s].encode_pframe(frames[s])
                assert np.array_equal(mv[s], omv), f"frame {t} stream {s}: motion vectors differ"
                assert np.array_equal(has[s], ohas), f"frame {t} stream {s}: skip flags differ"
                assert np.array_equal(coef[s], ocoef), f"frame {t} stream {s}: p-frame coefficients differ"
            stats["coded"] += int(has.sum())
            stats["mbs"] += has.size
        recon = enc.prev_frame()
        fb = dec.framebuffer()
        out = dec.get_frame()
        ctx.download(fused, fused_dev)
        for s in range(n_streams):
            oprev = oencs[s].prev_frame()
            assert np.array_equal(recon[s], oprev), f"frame {t} stream {s}: encoder reconstruction differs"
            assert np.array_equal(fb[s], oprev), f"frame {t} stream {s}: decoder framebuffer differs"
            # retframe = crop of the padded framebuffer (src/dec.rs:195-197)
            f = pkg.VideoFrame.from_packed(width, height, oprev, padded=True)
            crop = np.concatenate([f.plane_y.image()[:height, :width].reshape(-1),
                                   f.plane_u.image()[:height // 2, :width // 2].reshape(-1),
                                   f.plane_v.image()[:height // 2, :width // 2].reshape(-1)])
            assert np.array_equal(out[s], crop), f"frame {t} stream {s}: cropped retframe differs"
            assert np.array_equal(fused[s], crop), f"frame {t} stream {s}: fused retframe output differs"
    dec.set_output_dev(None)
    ctx.free(fused_dev)
    enc.close()
    dec.close()
    return stats


def check_session_batched_dev(pkg, ctx, oracle, width, height, quality, seeds, n_frames=2, gop=15, threads=1, kind="pan"):
    """The BENCHED shape (bench.py StreamSet.step): len(seeds) streams in ONE launch per frame operation through the
    device-pointer entry points, frames generated on the device, retframe crop fused into the decode kernels -- and then
    every byte of every stream (coefficients, motion vectors, skip flags, encoder reconstruction, decoder framebuffer,
    cropped retframe) against the oracle, stream by stream.  With 96 x 1080p the stream offsets reach 601 MB into the
    coefficient buffer."""
    S = len(seeds)
    enc = pkg.EncoderSession(ctx, width, height, quality, S)
    dec = pkg.DecoderSession(ctx, width, height, np.stack(pkg.qtables_from_quality(quality)[:4]), S)
    nb, fb, pfb = enc.total_blocks, enc.frame_bytes, enc.padded_frame_bytes
    d_frames = ctx.alloc(S * fb)
    d_coef, d_mv, d_has = ctx.alloc(S * nb * 512), ctx.alloc(S * nb * 2), ctx.alloc(S * nb)
    d_out = ctx.alloc(S * fb)
    dec.set_output_dev(d_out)
    oencs = [oracle.encoder(width, height, quality, threads) for _ in range(S)]
    frames = np.empty((S, fb), np.uint8)
    coef = np.empty((S, nb, 256), np.int16)
    mv, has = np.empty((S, nb, 2), np.int8), np.empty((S, nb), np.uint8)
    out = np.empty((S, fb), np.uint8)
    stats = {"coded": 0, "mbs": 0, "streams": S}
    for t in range(n_frames):
        ctx.synth_frames_dev(width, height, seeds, t, d_frames, kind=kind)
        if t % gop == 0:
            enc.encode_iframe_dev(d_frames, d_coef)
            dec.decode_iframe_dev(d_coef)
        else:
            enc.encode_pframe_dev(d_frames, d_mv, d_has, d_coef)
            dec.decode_pframe_dev(d_mv, d_has, d_coef)
        dec.check()
        ctx.download(frames, d_frames)
        ctx.download(coef, d_coef)
        ctx.download(out, d_out)
        if t % gop:
            ctx.download(mv, d_mv)
            ctx.download(has, d_has)
            stats["coded"] += int(has.sum())
            stats["mbs"] += has.size
        recon, fbuf = enc.prev_frame(), dec.framebuffer()
        for s in (0, S - 1):          # the device generator against synth.py for the first and the last stream
            assert np.array_equal(frames[s], pkg.SyntheticStream(width, height, seed=int(seeds[s]), kind=kind).frame(t)), f"frame {t} stream {s}: generator"
        for s in range(S):
            if t % gop == 0:
                ocoef = oencs[s].encode_iframe(frames[s])
            else:
                omv, ohas, ocoef = oencs[s].encode_pframe(frames[s])
                assert np.array_equal(mv[s], omv), f"frame {t} stream {s}: motion vectors differ"
                assert np.array_equal(has[s], ohas), f"frame {t} stream {s}: skip flags differ"
            assert np.array_equal(coef[s], ocoef), f"frame {t} stream {s}: coefficients differ"
            oprev = oencs[s].prev_frame()
            assert np.array_equal(recon[s], oprev), f"frame {t} stream {s}: encoder reconstruction differs"
            assert np.array_equal(fbuf[s], oprev), f"frame {t} stream {s}: decoder framebuffer differs"
            f = pkg.VideoFrame.from_packed(width, height, oprev, padded=True)
            crop = np.concatenate([f.plane_y.image()[:height, :width].reshape(-1),
                                   f.plane_u.image()[:height // 2, :width // 2].reshape(-1),
                                   f.plane_v.image()[:height // 2, :width // 2].reshape(-1)])
            assert np.array_equal(out[s], crop), f"frame {t} stream {s}: fused retframe differs"
    dec.set_output_dev(None)
    for p in (d_frames, d_coef, d_mv, d_has, d_out):
        ctx.free(p)
    enc.close()
    dec.close()
    return stats


def check_gop_batched_session(pkg, ctx, oracle, width, height, quality, n_frames, gop, seed=None, kind="pan", threads=1):
    """GOP-batched use of the sessions (include/pfv_hip.h, pfv_enc_session_set_window): ONE stream, frames resident on the device in
    display order, the slots of one encoder / decoder session hold its GOPs (an i-frame never reads prev_frame, src/enc.rs:84-97), frame
    t of every GOP in one launch per frame operation -- input stride = gop frames, a shorter last GOP through the slot window, device
    entropy stage on the window, decoded frames written back in display order (strided fused crop).  Every coefficient, motion vector,
    skip flag, packet payload byte and decoded pixel against the oracle's SERIAL encoder run over the same frames in display order."""
    import ctypes
    L = _oracle_serializers(oracle)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    seed = pkg.synth.SEED if seed is None else int(seed)
    n_gops = (n_frames + gop - 1) // gop
    enc = pkg.EncoderSession(ctx, width, height, quality, n_gops)
    dec = pkg.DecoderSession(ctx, width, height, np.stack(pkg.qtables_from_quality(quality)[:4]), n_gops)
    enc.enable_entropy()
    nb, fb = enc.total_blocks, enc.frame_bytes
    d_frames, d_out = ctx.alloc(n_frames * fb), ctx.alloc(n_frames * fb)
    for t in range(n_frames):
        ctx.synth_frames_dev(width, height, [seed], t, d_frames + t * fb, kind=kind)
    frames = np.empty((n_frames, fb), np.uint8)
    ctx.download(frames, d_frames)
    ctx.upload(d_out, np.full(n_frames * fb, 0xA5, np.uint8))
    d_coef, d_mv, d_has = ctx.alloc(n_gops * nb * 512), ctx.alloc(n_gops * nb * 2), ctx.alloc(n_gops * nb)
    enc.set_frame_stride(gop * fb)
    # the serial oracle over the display order
    oenc = oracle.encoder(width, height, quality, threads)
    want = []
    ref = np.zeros(int(ctx._lib.pfv_payload_worst_case(width, height)) + 64, np.uint8)
    for f in range(n_frames):
        if f % gop == 0:
            ocoef = oenc.encode_iframe(frames[f]); omv = ohas = None
            n = L.pfvo_serialize_iframe(P(ocoef), nb, P(ref), ref.size)
        else:
            omv, ohas, ocoef = oenc.encode_pframe(frames[f])
            n = L.pfvo_serialize_pframe(P(omv), P(ohas), P(ocoef), nb, P(ref), ref.size)
        pf = pkg.VideoFrame.from_packed(width, height, oenc.prev_frame(), padded=True)
        crop = np.concatenate([pf.plane_y.image()[:height, :width].reshape(-1), pf.plane_u.image()[:height // 2, :width // 2].reshape(-1),
                               pf.plane_v.image()[:height // 2, :width // 2].reshape(-1)])
        want.append((ocoef.copy(), None if omv is None else omv.copy(), None if ohas is None else ohas.copy(), ref[:n].tobytes(), crop))
    coef, mv, has = np.empty((n_gops, nb, 256), np.int16), np.empty((n_gops, nb, 2), np.int8), np.empty((n_gops, nb), np.uint8)
    launches = 0
    for t in range(min(gop, n_frames)):
        count = sum(1 for g in range(n_gops) if g * gop + t < n_frames)      # GOPs that have a frame t: a prefix (only the last may be short)
        enc.set_window(0, count)
        dec.set_window(0, count)
        dec.set_output_strided_dev(d_out + t * fb, gop * fb)
        if t == 0:
            enc.encode_iframe_dev(d_frames, d_coef)
            enc.pack_iframe_dev(d_coef)
            dec.decode_iframe_dev(d_coef)
        else:
            enc.encode_pframe_dev(d_frames + t * fb, d_mv, d_has, d_coef)
            enc.pack_pframe_dev(d_mv, d_has, d_coef)
            dec.decode_pframe_dev(d_mv, d_has, d_coef)
        launches += 1
        dec.check()
        sizes = enc.payload_sizes()
        ctx.download(coef, d_coef); ctx.download(mv, d_mv); ctx.download(has, d_has)
        for g in range(count):
            f = g * gop + t
            ocoef, omv, ohas, opay, _ = want[f]
            assert np.array_equal(coef[g], ocoef), f"frame {f} (GOP {g}, step {t}): coefficients differ from the serial oracle"
            if t:
                assert np.array_equal(mv[g], omv), f"frame {f}: motion vectors differ"
                assert np.array_equal(has[g], ohas), f"frame {f}: skip flags differ"
            assert int(sizes[g]) == len(opay) and enc.payload(g, len(opay)) == opay, f"frame {f}: packet payload differs from the serial oracle"
    out = np.empty((n_frames, fb), np.uint8)
    ctx.download(out, d_out)
    for f in range(n_frames):
        assert np.array_equal(out[f], want[f][4]), f"frame {f}: decoded frame (display order) differs from the serial oracle"
    # the host-buffer entry points refuse a partial window / a stride instead of guessing
    try:
        enc.encode_iframe(np.zeros((n_gops, fb), np.uint8))
        raise AssertionError("host-buffer encode accepted a strided session")
    except pkg.PfvError as e:
        assert e.code == pkg._lib.PFV_ERR_STATE
    for p in (d_frames, d_out, d_coef, d_mv, d_has):
        ctx.free(p)
    enc.close(); dec.close()
    return {"gops": n_gops, "launches_per_operation": launches, "frames": n_frames}


def check_gop_batched_clip(pkg, ctx, oracle, width, height, quality, n_frames, gop, seed=None, threads=1, dec_gops=None):
    """The launch shape `bench.py --workload config5` times, whole clip: ONE stream of n_frames, all its GOPs in the slots of a launch (frame t
    of every GOP per frame operation), against the oracle's SERIAL encoder over the same frames (src/enc.rs:84-97: an i-frame never reads
    prev_frame, so the GOPs are independent; README.md:34-41 GOP pattern).  Like check_gop_batched_session, but sized for 300 4K frames:
    what the device produces per frame -- coefficients, motion vectors, skip flags, device-built packet payload, display-order decoded frame
    -- is kept as 128-bit BLAKE2 digests and compared with the digests of the oracle's output frame by frame (7.5 GB of coefficients need
    not sit in memory twice).  Then the packets, as a .pfv stream, through pfv_gop_decoder with the payloads read by the device's entropy
    stage, dec_gops GOPs per batch: every frame it delivers against the oracle's reconstruction."""
    import ctypes
    import hashlib
    dig = lambda a: hashlib.blake2b(np.ascontiguousarray(a), digest_size=16).digest()
    L = _oracle_serializers(oracle)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    seed = pkg.synth.SEED if seed is None else int(seed)
    n_gops = (n_frames + gop - 1) // gop
    enc = pkg.EncoderSession(ctx, width, height, quality, n_gops)
    tabs = pkg.qtables_from_quality(quality)
    dec = pkg.DecoderSession(ctx, width, height, np.stack(tabs[:4]), n_gops)
    enc.enable_entropy()
    nb, fb = enc.total_blocks, enc.frame_bytes
    d_frames, d_out = ctx.alloc(n_frames * fb), ctx.alloc(n_frames * fb)
    for t in range(n_frames):
        ctx.synth_frames_dev(width, height, [seed], t, d_frames + t * fb)
    d_coef, d_mv, d_has = ctx.alloc(n_gops * nb * 512), ctx.alloc(n_gops * nb * 2), ctx.alloc(n_gops * nb)
    enc.set_frame_stride(gop * fb)
    coef, mv, has = np.empty((n_gops, nb, 256), np.int16), np.empty((n_gops, nb, 2), np.int8), np.empty((n_gops, nb), np.uint8)
    got, payloads = [None] * n_frames, [None] * n_frames
    for t in range(min(gop, n_frames)):
        count = sum(1 for g in range(n_gops) if g * gop + t < n_frames)
        enc.set_window(0, count)
        dec.set_window(0, count)
        dec.set_output_strided_dev(d_out + t * fb, gop * fb)
        if t == 0:
            enc.encode_iframe_dev(d_frames, d_coef)
            enc.pack_iframe_dev(d_coef)
            dec.decode_iframe_dev(d_coef)
        else:
            enc.encode_pframe_dev(d_frames + t * fb, d_mv, d_has, d_coef)
            enc.pack_pframe_dev(d_mv, d_has, d_coef)
            dec.decode_pframe_dev(d_mv, d_has, d_coef)
        dec.check()
        sizes = enc.payload_sizes()
        ctx.download(coef, d_coef); ctx.download(mv, d_mv); ctx.download(has, d_has)
        for g in range(count):
            f = g * gop + t
            payloads[f] = enc.payload(g, int(sizes[g]))
            got[f] = (dig(coef[g]), dig(mv[g]) if t else None, dig(has[g]) if t else None, hashlib.blake2b(payloads[f], digest_size=16).digest())
    out = np.empty(fb, np.uint8)
    frame = np.empty(fb, np.uint8)
    oenc = oracle.encoder(width, height, quality, threads)
    ref = np.zeros(int(ctx._lib.pfv_payload_worst_case(width, height)) + 64, np.uint8)
    want_frames = []
    for f in range(n_frames):
        ctx.download(frame, d_frames + f * fb)
        if f % gop == 0:
            ocoef = oenc.encode_iframe(frame); omv = ohas = None
            n = L.pfvo_serialize_iframe(P(ocoef), nb, P(ref), ref.size)
        else:
            omv, ohas, ocoef = oenc.encode_pframe(frame)
            n = L.pfvo_serialize_pframe(P(omv), P(ohas), P(ocoef), nb, P(ref), ref.size)
        pf = pkg.VideoFrame.from_packed(width, height, oenc.prev_frame(), padded=True)
        crop = np.concatenate([pf.plane_y.image()[:height, :width].reshape(-1), pf.plane_u.image()[:height // 2, :width // 2].reshape(-1),
                               pf.plane_v.image()[:height // 2, :width // 2].reshape(-1)])
        g = got[f]
        assert g[0] == dig(ocoef), f"frame {f}: coefficients differ from the serial oracle"
        if f % gop:
            assert g[1] == dig(omv), f"frame {f}: motion vectors differ"
            assert g[2] == dig(ohas), f"frame {f}: skip flags differ"
        assert len(payloads[f]) == n and g[3] == hashlib.blake2b(ref[:n].tobytes(), digest_size=16).digest(), f"frame {f}: packet payload differs from the serial oracle"
        ctx.download(out, d_out + f * fb)
        assert np.array_equal(out, crop), f"frame {f}: decoded frame (display order) differs from the serial oracle"
        want_frames.append(dig(crop))
    for p in (d_frames, d_out, d_coef, d_mv, d_has):
        ctx.free(p)
    enc.close(); dec.close()
    # the same packets as a .pfv stream (src/enc.rs:190-235: magic, version, geometry, the four q-tables; type:u8 len:u32 payload) through the
    # GOP-batched decoder object, payloads read on the device
    head = b"PFVIDEO\0" + (211).to_bytes(4, "little") + b"".join(int(v).to_bytes(2, "little") for v in (width, height, 30, 4))
    head += b"".join(np.asarray(tabs[k], dtype="<u2").tobytes() for k in range(4))
    stream = head + b"".join(bytes([1 if f % gop == 0 else 2]) + len(payloads[f]).to_bytes(4, "little") + payloads[f] for f in range(n_frames)) + bytes(5)
    gdec = pkg.GopDecoder(stream, ctx, max_gops=dec_gops or n_gops, max_gop_frames=gop, threads=max(2, min(threads, 15)), raw=True, entropy="device")
    n_got = [0]

    def onvideo(y, u, v):
        f = n_got[0]
        assert hashlib.blake2b(np.concatenate([y, u, v]), digest_size=16).digest() == want_frames[f], f"frame {f}: pfv_gop_decoder (device entropy) differs from the oracle"
        n_got[0] += 1
    while gdec.advance_frame(onvideo):
        pass
    stats = gdec.stats()
    gdec.close()
    assert n_got[0] == n_frames
    return {"gops": n_gops, "frames": n_frames, "stream_bytes": len(stream), "packets_read_on_device": stats["packets_read_on_device"],
            "packets_left_to_host_parser": stats["packets_left_to_host_parser"]}


def check_golden(pkg, ctx, oracle):
    """the HIP path against the committed known-answer vectors (tests/golden/hotpath_vectors.npz)"""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_vectors.npz"))
    il, _, pl, _, px_err = oracle.qtables(5)
    # the src/lib.rs:61-66 block as the top-left subblock of a 16x16 plane (rest = clear colour 128 -> zero AC)
    plane = pkg.VideoPlane.from_slice(8, 8, gold["lib_block"])
    enc = plane.encode_plane(gold["lib_q"], 128, ctx)
    assert np.array_equal(enc.blocks[0, :64], gold["lib_quant"])
    rec = pkg.VideoPlane.decode_plane(enc, gold["lib_q"], ctx)
    assert np.array_equal(rec.image()[:8, :8].reshape(-1), gold["lib_recon"])
    # 64 random subblocks, 4 per macroblock, all golden qualities / tables
    px = gold["sub_px"].reshape(16, 4, 8, 8)
    img = px.reshape(16, 2, 2, 8, 8).transpose(0, 1, 3, 2, 4).reshape(16 * 16, 16)     # 16 macroblocks stacked vertically
    plane = pkg.VideoPlane.from_slice(16, 256, img)
    for quality in (0, 2, 5, 10):
        tabs = oracle.qtables(quality)
        for name, q in (("intra_l", tabs[0]), ("intra_c", tabs[1]), ("inter_l", tabs[2])):
            e = plane.encode_plane(q, 0, ctx)
            assert np.array_equal(e.blocks.reshape(64, 64), gold[f"q{quality}_{name}_enc"]), (quality, name)
            d = pkg.VideoPlane.decode_plane(e, q, ctx).image().reshape(16, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(64, 64)
            assert np.array_equal(d, gold[f"q{quality}_{name}_dec"]), (quality, name)
    # the 64x48 p-frame case
    f0 = pkg.VideoPlane.from_slice(64, 48, gold["pf_f0"])
    f1 = pkg.VideoPlane.from_slice(64, 48, gold["pf_f1"])
    e0 = f0.encode_plane(il, 0, ctx)
    assert np.array_equal(e0.blocks, gold["pf_c0"])
    r0 = pkg.VideoPlane.decode_plane(e0, il, ctx)
    assert np.array_equal(r0.image(), gold["pf_rec0"])
    e1 = f1.encode_plane_delta(r0, pl, px_err, 0, ctx)
    assert np.array_equal(e1.motion, gold["pf_mv"]) and np.array_equal(e1.has_coeff, gold["pf_has"])
    assert np.array_equal(e1.blocks, gold["pf_c1"])
    r1 = pkg.VideoPlane.decode_plane_delta(e1, r0, pl, ctx)
    assert np.array_equal(r1.image(), gold["pf_rec1"])


def check_trap_vectors(pkg, ctx, oracle):
    """the HIP path against tests/golden/trap_vectors.npz: the vectors aimed at the bit-exactness traps of SURVEY.md section 8c
    (pad colour on a ragged plane, exact ties, skip threshold met with equality, last legal search position, i32 wrap in
    decode); tests/test_mutation_sensitivity.py shows which rule each of them notices"""
    import os
    t = np.load(os.path.join(os.path.dirname(__file__), "golden", "trap_vectors.npz"))
    _, ic, pl, pcq, px_err = oracle.qtables(5)
    f0 = pkg.VideoPlane.from_slice(50, 38, t["rag_f0"])
    e0 = f0.encode_plane(ic, 128, ctx)
    assert np.array_equal(e0.blocks, t["rag_c0"])
    r0 = pkg.VideoPlane.decode_plane(e0, ic, ctx)
    assert np.array_equal(r0.image(), t["rag_rec0"])
    e1 = pkg.VideoPlane.from_slice(50, 38, t["rag_f1"]).encode_plane_delta(r0, pcq, px_err, 128, ctx)
    assert np.array_equal(e1.motion, t["rag_mv"]) and np.array_equal(e1.has_coeff, t["rag_has"]) and np.array_equal(e1.blocks, t["rag_c1"])
    assert np.array_equal(pkg.VideoPlane.decode_plane_delta(e1, r0, pcq, ctx).image(), t["rag_rec1"])
    for name in ("tie", "diag", "edge"):
        src, ref = t[f"{name}_src"], t[f"{name}_ref"]
        h, w = src.shape
        e = pkg.VideoPlane.from_slice(w, h, src).encode_plane_delta(pkg.VideoPlane.from_slice(w, h, ref), pl, px_err, 0, ctx)
        assert np.array_equal(e.motion, t[f"{name}_mv"]), name
        assert np.array_equal(e.has_coeff, t[f"{name}_has"]), name
        assert np.array_equal(e.blocks, t[f"{name}_c1"]), name
    enc = pkg.EncodedIPlane(32, 64, 2, 4, t["host_coef"])
    rec = pkg.VideoPlane.decode_plane(enc, t["host_q"], ctx).image()
    want = t["host_rec"].reshape(4, 2, 16, 16).transpose(0, 2, 1, 3).reshape(64, 32)
    assert np.array_equal(rec, want)


def check_colour_utils(pkg, ctx, oracle):
    """pfv_reduce_dev / pfv_double_dev against the ORACLE's VideoPlane::reduce / double (src/common.rs:523-556,
    oracle/pfv_oracle.c pfvo_reduce / pfvo_double); the package's own numpy mirror is checked against the oracle too"""
    import ctypes
    rng = np.random.default_rng(8)
    for (w, h) in [(37, 21), (64, 48), (2, 2), (1, 1), (3, 1), (1, 3), (640, 360), (1921, 1081), (5, 4096)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        src = pkg.VideoPlane.from_slice(w, h, img)
        d_src = ctx.alloc(max(w * h, 1))
        ctx.upload(d_src, src.pixels)
        want = oracle.reduce(img)
        assert np.array_equal(src.reduce().image(), want), ("host mirror reduce", w, h)
        if want.size:
            d_red = ctx.alloc(want.size + 16)
            guard = np.full(want.size + 16, 0xA5, np.uint8)
            ctx.upload(d_red, guard)
            ctx.check(ctx._lib.pfv_reduce_dev(ctx.handle, ctypes.c_void_p(d_red), ctypes.c_void_p(d_src), w, h))
            out = np.empty(want.size + 16, np.uint8)
            ctx.download(out, d_red)
            assert np.array_equal(out[:want.size], want.reshape(-1)), ("reduce", w, h)
            assert (out[want.size:] == 0xA5).all(), ("reduce wrote past its plane", w, h)
            ctx.free(d_red)
        want = oracle.double(img)
        assert np.array_equal(src.double().image(), want), ("host mirror double", w, h)
        d_dbl = ctx.alloc(want.size + 16)
        guard = np.full(want.size + 16, 0x5A, np.uint8)
        ctx.upload(d_dbl, guard)
        ctx.check(ctx._lib.pfv_double_dev(ctx.handle, ctypes.c_void_p(d_dbl), ctypes.c_void_p(d_src), w, h))
        out = np.empty(want.size + 16, np.uint8)
        ctx.download(out, d_dbl)
        assert np.array_equal(out[:want.size], want.reshape(-1)), ("double", w, h)
        assert (out[want.size:] == 0x5A).all(), ("double wrote past its plane", w, h)
        ctx.free(d_dbl)
        ctx.free(d_src)


def check_blit_dev(pkg, ctx, oracle, n_random=64, seed=5):
    """pfv_blit_dev against the ORACLE's VideoPlane::blit (src/plane.rs:20-29, pfvo_blit): random rectangles between
    planes of unrelated widths plus the corner cases -- one pixel, one row, one column, the full plane, rectangles that
    touch every edge of source and destination, unaligned origins and widths.  Everything outside the rectangle must
    keep its old value.  The package's numpy mirror (VideoPlane.blit) is held to the same oracle."""
    import ctypes
    rng = np.random.default_rng(seed)
    geoms = [((100, 40), (64, 64)), ((1920, 1080), (1920, 1088)), ((17, 33), (33, 17)), ((1, 1), (1, 1)), ((255, 3), (300, 7))]
    n_checked = 0
    for (sw_, sh_), (dw_, dh_) in geoms:
        simg = rng.integers(0, 256, (sh_, sw_), dtype=np.uint8)
        dimg = rng.integers(0, 256, (dh_, dw_), dtype=np.uint8)
        d_src, d_dst = ctx.alloc(simg.size), ctx.alloc(dimg.size)
        ctx.upload(d_src, simg)
        mw, mh = min(sw_, dw_), min(sh_, dh_)
        rects = [(0, 0, 0, 0, 1, 1),                                   # one pixel, origin
                 (dw_ - 1, dh_ - 1, sw_ - 1, sh_ - 1, 1, 1),           # one pixel, last corner of both
                 (0, 0, 0, 0, mw, mh),                                 # the largest common rectangle, top left
                 (dw_ - mw, dh_ - mh, sw_ - mw, sh_ - mh, mw, mh),     # ... bottom right: touches right / bottom edges
                 (0, dh_ - 1, 0, 0, mw, 1),                            # one row into the last row
                 (dw_ - 1, 0, 0, 0, 1, mh),                            # one column into the last column
                 (0, 0, sw_ - 1, 0, 1, mh)]                            # last source column
        if sw_ == dw_ and sh_ <= dh_:
            rects.append((0, 0, 0, 0, sw_, sh_))                       # full-plane copy (the pad blit of common.rs:356)
        for _ in range(n_random if mw > 1 else 2):
            w = int(rng.integers(1, mw + 1)); h = int(rng.integers(1, mh + 1))
            rects.append((int(rng.integers(0, dw_ - w + 1)), int(rng.integers(0, dh_ - h + 1)),
                          int(rng.integers(0, sw_ - w + 1)), int(rng.integers(0, sh_ - h + 1)), w, h))
        for (dx, dy, sx, sy, w, h) in rects:
            ctx.upload(d_dst, dimg)
            ctx.check(ctx._lib.pfv_blit_dev(ctx.handle, ctypes.c_void_p(d_dst), dw_, dh_, ctypes.c_void_p(d_src), sw_, sh_,
                                            dx, dy, sx, sy, w, h))
            out = np.empty(dimg.shape, np.uint8)
            ctx.download(out, d_dst)
            want = oracle.blit(dimg, simg, dx, dy, sx, sy, w, h)
            assert np.array_equal(out, want), ("pfv_blit_dev", (sw_, sh_), (dw_, dh_), (dx, dy, sx, sy, w, h))
            n_checked += 1
        # the host mirror, a few rectangles per geometry
        for (dx, dy, sx, sy, w, h) in rects[:12]:
            dst = pkg.VideoPlane.from_slice(dw_, dh_, dimg)
            dst.blit(pkg.VideoPlane.from_slice(sw_, sh_, simg), dx, dy, sx, sy, w, h)
            assert np.array_equal(dst.image(), oracle.blit(dimg, simg, dx, dy, sx, sy, w, h)), "host mirror blit"
        # empty rectangles are legal no-ops (0..0 loops in the reference); rectangles outside either plane are errors
        ctx.upload(d_dst, dimg)
        ctx.check(ctx._lib.pfv_blit_dev(ctx.handle, ctypes.c_void_p(d_dst), dw_, dh_, ctypes.c_void_p(d_src), sw_, sh_, 0, 0, 0, 0, 0, 1))
        ctx.check(ctx._lib.pfv_blit_dev(ctx.handle, ctypes.c_void_p(d_dst), dw_, dh_, ctypes.c_void_p(d_src), sw_, sh_, 0, 0, 0, 0, 1, 0))
        out = np.empty(dimg.shape, np.uint8)
        ctx.download(out, d_dst)
        assert np.array_equal(out, dimg)
        for bad in ((dw_, 0, 0, 0, 1, 1), (0, dh_, 0, 0, 1, 1), (0, 0, sw_, 0, 1, 1), (0, 0, 0, sh_, 1, 1), (0, 0, 0, 0, mw + max(sw_, dw_), 1),
                    (-1, 0, 0, 0, 1, 1), (0, 0, 0, -1, 1, 1)):
            rc = ctx._lib.pfv_blit_dev(ctx.handle, ctypes.c_void_p(d_dst), dw_, dh_, ctypes.c_void_p(d_src), sw_, sh_, *bad)
            assert rc == pkg._lib.PFV_ERR_BAD_ARG, bad
        ctx.free(d_src); ctx.free(d_dst)
    return n_checked


def check_misaligned_device_frames(pkg, ctx, oracle):
    """device-pointer session entry points with a frame buffer that is NOT 16-byte aligned (byte-load fallback of
    load_src16 / separate crop pass instead of the fused one)"""
    w, h, q = 64, 48, 5
    st = pkg.SyntheticStream(w, h)
    f0, f1 = st.frame(0), st.frame(1)
    enc = pkg.EncoderSession(ctx, w, h, q, 1)
    dec = pkg.DecoderSession(ctx, w, h, np.stack(pkg.qtables_from_quality(q)[:4]), 1)
    nb, fb = enc.total_blocks, enc.frame_bytes
    d_frames = ctx.alloc(fb + 64)
    d_out = ctx.alloc(fb + 64)
    d_coef, d_mv, d_has = ctx.alloc(nb * 512), ctx.alloc(nb * 2), ctx.alloc(nb)
    oenc = oracle.encoder(w, h, q)
    dec.set_output_dev(d_out + 3)                      # misaligned retframe target
    for t, f in enumerate((f0, f1)):
        ctx.upload(d_frames + 1, f)                    # misaligned source frames
        if t == 0:
            enc.encode_iframe_dev(d_frames + 1, d_coef)
            dec.decode_iframe_dev(d_coef)
            ocoef = oenc.encode_iframe(f)
        else:
            enc.encode_pframe_dev(d_frames + 1, d_mv, d_has, d_coef)
            dec.decode_pframe_dev(d_mv, d_has, d_coef)
            omv, ohas, ocoef = oenc.encode_pframe(f)
            mv, has = np.empty((nb, 2), np.int8), np.empty(nb, np.uint8)
            ctx.download(mv, d_mv); ctx.download(has, d_has)
            assert np.array_equal(mv, omv) and np.array_equal(has, ohas)
        coef = np.empty((nb, 256), np.int16)
        ctx.download(coef, d_coef)
        assert np.array_equal(coef, ocoef)
        dec.check()
        assert np.array_equal(enc.prev_frame()[0], oenc.prev_frame())
        out = np.empty(fb, np.uint8)
        ctx.download(out, d_out + 3)
        assert np.array_equal(out, dec.get_frame()[0])
    dec.set_output_dev(None)
    for p in (d_frames, d_out, d_coef, d_mv, d_has):
        ctx.free(p)
    enc.close(); dec.close()


def fuzz_plane_ops(pkg, ctx, oracle, n_cases, seed, max_w=420, max_h=200):
    """randomised geometry / content / quantiser / skip-threshold sweep of the four plane operators"""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        w, h = int(rng.integers(1, max_w + 1)), int(rng.integers(1, max_h + 1))
        kind = case % 4
        if kind == 0:
            px = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            px = smooth_plane(h, w, seed * 1000 + case)
        elif kind == 2:
            px = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
        else:
            px = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        q = rng.integers(1, int(rng.choice([4, 40, 400])), 64).astype(np.int32)
        clear = int(rng.integers(0, 256))
        enc, dec = check_encode_plane(pkg, ctx, oracle, px, q, clear)
        ref = dec.image().copy()
        ref = np.ascontiguousarray(np.roll(ref, (int(rng.integers(-15, 16)), int(rng.integers(-15, 16))), (0, 1)))
        if case % 3 == 0:
            ref = np.clip(ref.astype(int) + rng.integers(-20, 21, ref.shape), 0, 255).astype(np.uint8)
        px_err = float(rng.choice([0.0, 1.5, 7.5, 15.0]))
        check_encode_plane_delta(pkg, ctx, oracle, px, ref, q, px_err, clear)


def _oracle_serializers(oracle):
    import ctypes
    L = oracle.L
    L.pfvo_serialize_iframe.restype = ctypes.c_size_t
    L.pfvo_serialize_iframe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    L.pfvo_serialize_pframe.restype = ctypes.c_size_t
    L.pfvo_serialize_pframe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    return L


def _hostile_coefficients(rng, nb, kind):
    """coefficient buffers that stress the run coder: dense, sparse, long runs, lone last value, every size class"""
    if kind == "zero":
        return np.zeros((nb, 256), np.int16)
    if kind == "dense":
        c = rng.integers(-16383, 16384, (nb, 256))
        c[c == 0] = 1
        return c.astype(np.int16)
    if kind == "sparse":
        return (rng.integers(-300, 301, (nb, 256)) * (rng.random((nb, 256)) < 0.03)).astype(np.int16)
    if kind == "typical":       # low-frequency-heavy, like quantised DCT output in zigzag order
        keep = rng.random((nb, 256)) < (0.6 * np.exp(-(np.arange(256) % 64) / 6.0))[None, :]
        return (rng.integers(-40, 41, (nb, 256)) * keep).astype(np.int16)
    if kind == "edges":
        c = np.zeros((nb, 256), np.int16)
        for b in range(nb):
            pick = int(rng.integers(0, 8))
            if pick == 0: c[b, 255] = -1                                   # 255 zeros, then one value
            elif pick == 1: c[b, 0] = 16383                                # size 15, then 255 zeros
            elif pick == 2: c[b, ::16] = rng.integers(1, 9, 16)            # runs of exactly 15
            elif pick == 3: c[b, 16::17] = -2                              # runs of 16 (one filler each)
            elif pick == 4: c[b, [63, 64, 127, 128, 191, 192]] = 7         # values on subblock boundaries
            elif pick == 5: c[b, 31] = -16383; c[b, 62] = 1                # runs of 31 and 30
            elif pick == 6: c[b, 64 * int(rng.integers(0, 4)) + int(rng.integers(0, 64))] = int(rng.integers(-9, 10))
            # pick == 7: all zero
        return c
    raise ValueError(kind)


def check_device_entropy(pkg, ctx, oracle, w, h, n_streams, seed, kinds=("typical", "zero", "dense", "sparse", "edges")):
    """the device entropy stage (k_ent_*) on arbitrary coefficient / header buffers: payload bytes identical to the
    oracle's write_iframe_packet / write_pframe_packet restatement, stream by stream"""
    import ctypes
    L = _oracle_serializers(oracle)
    rng = np.random.default_rng(seed)
    enc = pkg.EncoderSession(ctx, w, h, 5, n_streams)
    enc.enable_entropy()
    nb, S = enc.total_blocks, n_streams
    d_coef, d_mv, d_has = ctx.alloc(S * nb * 512), ctx.alloc(S * nb * 2), ctx.alloc(S * nb)
    cap = int(ctx._lib.pfv_payload_worst_case(w, h))
    ref = np.zeros(cap + 64, np.uint8)
    checked = 0
    for kind in kinds:
        coef = np.stack([_hostile_coefficients(rng, nb, kind if s % 2 == 0 else "typical") for s in range(S)])
        mv = rng.integers(-15, 16, (S, nb, 2)).astype(np.int8)
        mv[:, ::3] = 0
        has = (rng.random((S, nb)) < 0.6).astype(np.uint8)
        if kind == "zero":
            has[0] = 0                                                     # a p-frame with no coded macroblock at all
        ctx.upload(d_coef, coef); ctx.upload(d_mv, mv); ctx.upload(d_has, has)
        for pframe in (False, True):
            if pframe:
                enc.pack_pframe_dev(d_mv, d_has, d_coef)
            else:
                enc.pack_iframe_dev(d_coef)
            sizes = enc.payload_sizes()
            for s in range(S):
                if pframe:
                    n = L.pfvo_serialize_pframe(mv[s].ctypes.data_as(ctypes.c_void_p), has[s].ctypes.data_as(ctypes.c_void_p),
                                                coef[s].ctypes.data_as(ctypes.c_void_p), nb, ref.ctypes.data_as(ctypes.c_void_p), ref.size)
                else:
                    n = L.pfvo_serialize_iframe(coef[s].ctypes.data_as(ctypes.c_void_p), nb, ref.ctypes.data_as(ctypes.c_void_p), ref.size)
                assert n > 0 and int(sizes[s]) == n, (kind, pframe, s, int(sizes[s]), n)
                got = np.frombuffer(enc.payload(s, n), np.uint8)
                if not np.array_equal(got, ref[:n]):
                    bad = int(np.flatnonzero(got != ref[:n])[0])
                    raise AssertionError(f"{kind} pframe={pframe} stream {s}: payload differs at byte {bad} of {n}")
                checked += 1
    # a coefficient that needs 16 size bits: the reference panics (rle.rs:44); the stage reports it
    coef = np.zeros((S, nb, 256), np.int16)
    coef[S - 1, nb - 1, 200] = -16384
    ctx.upload(d_coef, coef)
    enc.pack_iframe_dev(d_coef)
    try:
        enc.payload_sizes()
        raise AssertionError("oversized coefficient not reported")
    except pkg.PfvError as e:
        assert e.code == pkg._lib.PFV_ERR_FORMAT
    coef[S - 1, nb - 1, 200] = 16383                                        # and the stage recovers on the next frame
    ctx.upload(d_coef, coef)
    enc.pack_iframe_dev(d_coef)
    assert int(enc.payload_sizes()[S - 1]) > 19
    for p in (d_coef, d_mv, d_has):
        ctx.free(p)
    enc.close()
    return checked


def check_sparse_decode(pkg, ctx, w=100, h=60, n_streams=2, seed=11):
    """pfv_dec_*_sparse == pfv_dec_* on the expanded coefficient array (i-frame then p-frame, two sessions side by side)"""
    rng = np.random.default_rng(seed)
    q = np.stack(pkg.qtables_from_quality(5)[:4])
    a = pkg.DecoderSession(ctx, w, h, q, n_streams)
    b = pkg.DecoderSession(ctx, w, h, q, n_streams)
    nb, S = a.total_blocks, n_streams
    for frame in range(3):
        kind = ("typical", "sparse", "zero")[frame]
        coef = np.stack([_hostile_coefficients(rng, nb, kind) for _ in range(S)])
        flat = coef.reshape(-1)
        idx = np.flatnonzero(flat).astype(np.uint32)
        val = flat[idx]
        # entries the dense form cannot express must not matter: an explicit zero, an index past the frame
        idx2 = np.concatenate([idx, np.array([flat.size + 5], np.uint32)])
        val2 = np.concatenate([val, np.array([77], np.int16)])
        if frame == 0:
            a.decode_iframe(coef)
            b.decode_iframe_sparse(idx2, val2)
        else:
            mv = np.zeros((S, nb, 2), np.int8)                 # zero motion is legal for every macroblock
            has = (rng.random((S, nb)) < 0.7).astype(np.uint8)
            a.decode_pframe(mv, has, coef)
            b.decode_pframe_sparse(mv, has, idx2, val2)
        assert np.array_equal(a.framebuffer(), b.framebuffer()), f"sparse != dense decode, frame {frame}"
    a.close(); b.close()


def check_lists_decode(pkg, ctx, w=100, h=60, n_streams=2, seed=13, kinds=("typical", "sparse", "zero", "dense", "edges")):
    """pfv_dec_*_lists_dev (coefficient lists expanded in the kernels' LDS stage) == pfv_dec_* on the dense array: an i-frame, then
    p-frames with every third macroblock or so skipped, two sessions side by side; the framebuffer after every frame"""
    rng = np.random.default_rng(seed)
    q = np.stack(pkg.qtables_from_quality(5)[:4])
    a = pkg.DecoderSession(ctx, w, h, q, n_streams)
    b = pkg.DecoderSession(ctx, w, h, q, n_streams)
    nb, S = a.total_blocks, n_streams
    n_entries = 0
    for frame, kind in enumerate(kinds):
        coef = np.stack([_hostile_coefficients(rng, nb, kind) for _ in range(S)])
        if frame == 0:
            entries, counts = b.coef_lists(coef)
            a.decode_iframe(coef)
            b.decode_iframe_lists(entries, counts)
        else:
            mv = np.zeros((S, nb, 2), np.int8)                 # zero motion is legal for every macroblock
            has = (rng.random((S, nb)) < 0.7).astype(np.uint8)
            if kind == "zero":
                has[:] = 1                                      # coded macroblocks without a single value: they own no entries
            entries, counts = b.coef_lists(coef, has)
            # the macroblocks a p-frame skips own no entries, whatever their coefficients say
            assert sum(e.size for e in entries) == int(np.count_nonzero(coef.reshape(S, nb, 256)[has.astype(bool)]))
            a.decode_pframe(mv, has, coef)
            b.decode_pframe_lists(mv, has, entries, counts)
        n_entries += sum(e.size for e in entries)
        assert np.array_equal(a.framebuffer(), b.framebuffer()), f"lists != dense decode, frame {frame} ({kind})"
    a.close(); b.close()
    return n_entries


def check_async_entropy(pkg, ctx, oracle, w, h, n_streams, n_frames=6):
    """entropy stage on its own HIP stream with two alternating sets of encode outputs: payloads equal the oracle's
    serialisation of the same encode outputs, frame by frame"""
    import ctypes
    L = _oracle_serializers(oracle)
    S = n_streams
    enc = pkg.EncoderSession(ctx, w, h, 5, S)
    enc.enable_entropy(async_stream=True)
    nb, fb = enc.total_blocks, enc.frame_bytes
    streams = [pkg.SyntheticStream(w, h, seed=pkg.synth.SEED + s) for s in range(S)]
    d_frames = [ctx.alloc(S * fb) for _ in range(2)]
    sets = [(ctx.alloc(S * nb * 512), ctx.alloc(S * nb * 2), ctx.alloc(S * nb)) for _ in range(2)]
    cap = int(ctx._lib.pfv_payload_worst_case(w, h))
    ref = np.zeros(cap + 64, np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for t in range(n_frames):
        d_c, d_m, d_h = sets[t & 1]
        ctx.upload(d_frames[t & 1], np.concatenate([st.frame(t) for st in streams]))
        if t % 4 == 0:
            enc.encode_iframe_dev(d_frames[t & 1], d_c)
            enc.pack_iframe_dev(d_c)
        else:
            enc.encode_pframe_dev(d_frames[t & 1], d_m, d_h, d_c)
            enc.pack_pframe_dev(d_m, d_h, d_c)
        sizes = enc.payload_sizes()                       # synchronises with the entropy stream only
        payloads = [enc.payload(s, int(sizes[s])) for s in range(S)]
        ctx.sync()
        coef, mv, has = np.empty((S, nb, 256), np.int16), np.empty((S, nb, 2), np.int8), np.empty((S, nb), np.uint8)
        ctx.download(coef, d_c)
        if t % 4:
            ctx.download(mv, d_m); ctx.download(has, d_h)
        for s in range(S):
            n = (L.pfvo_serialize_pframe(P(mv[s]), P(has[s]), P(coef[s]), nb, P(ref), ref.size) if t % 4 else
                 L.pfvo_serialize_iframe(P(coef[s]), nb, P(ref), ref.size))
            assert n == int(sizes[s]) and payloads[s] == ref[:n].tobytes(), f"frame {t} stream {s}"
    enc.entropy_join()
    ctx.sync()
    for p in d_frames + [x for st in sets for x in st]:
        ctx.free(p)
    enc.close()


def _oracle_colour(oracle):
    import ctypes
    L = oracle.L
    L.pfvo_rgb_to_yuv420.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.pfvo_yuv420_to_rgb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.pfvo_rgb_to_yuv420.restype = L.pfvo_yuv420_to_rgb.restype = None
    return L


def all_rgb_image():
    """8192 x 8192 RGB image in which every one of the 2^24 colours fills one 2x2 block (so each is chroma-sampled)"""
    idx = np.arange(1 << 24, dtype=np.uint32).reshape(4096, 4096)
    big = np.repeat(np.repeat(idx, 2, axis=0), 2, axis=1)
    return np.stack([(big & 255).astype(np.uint8), ((big >> 8) & 255).astype(np.uint8), (big >> 16).astype(np.uint8)], axis=-1)


def all_yuv_frame():
    """4096 x 4096 4:2:0 frame that contains every (Y, U, V) triple"""
    w = h = 4096
    x, y = np.meshgrid(np.arange(w, dtype=np.int32), np.arange(h, dtype=np.int32))
    cx, cy = x >> 1, y >> 1
    Y = (((cx >> 8) + 8 * (cy >> 8)) * 4 + (x & 1) + 2 * (y & 1)).astype(np.uint8)
    c = np.arange(2048, dtype=np.int32)
    U = np.broadcast_to((c & 255).astype(np.uint8)[None, :], (2048, 2048))
    V = np.broadcast_to((c & 255).astype(np.uint8)[:, None], (2048, 2048))
    return np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)]), w, h


def check_colour_conversions(pkg, ctx, oracle, exhaustive):
    """pfv_rgb_to_yuv420_dev / pfv_yuv420_to_rgb_dev == the oracle's restatement of the reference's test helpers"""
    import ctypes
    L = _oracle_colour(oracle)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(7)
    cases = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for (w, h) in ((2, 2), (34, 18), (640, 360))]
    if exhaustive:
        cases.append(all_rgb_image())
    for rgb in cases:
        h, w = rgb.shape[:2]
        ref = np.empty(w * h + 2 * (w // 2) * (h // 2), np.uint8)
        L.pfvo_rgb_to_yuv420(P(rgb), w, h, P(ref))
        fr = pkg.VideoFrame.from_rgb(ctx, rgb)
        assert np.array_equal(fr.packed(), ref), f"rgb -> yuv420 differs ({w}x{h})"
    frames = [(rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8), w, h) for (w, h) in ((2, 2), (34, 18), (640, 360))]
    if exhaustive:
        frames.append(all_yuv_frame())
    for buf, w, h in frames:
        ref = np.empty((h, w, 3), np.uint8)
        L.pfvo_yuv420_to_rgb(P(buf), w, h, P(ref))
        got = pkg.VideoFrame.from_packed(w, h, buf).to_rgb(ctx)
        assert np.array_equal(got, ref), f"yuv420 -> rgb differs ({w}x{h})"


def check_gop_graph(pkg, ctx, oracle, w=64, h=48, n_streams=2, n_frames=4, quality=5):
    """A GOP recorded once as a HIP graph (pfv_graph_begin / _end) and replayed: the buffers after every replay equal the
    oracle's for the same frames -- the replay really re-runs encode + decode, and the sessions' ping-pong state is consistent."""
    fb = int(pkg._lib.load().pfv_frame_bytes(w, h))
    seeds = [pkg.synth.SEED + 5 * k for k in range(n_streams)]
    enc = pkg.EncoderSession(ctx, w, h, quality, n_streams)
    dec = pkg.DecoderSession(ctx, w, h, np.stack(pkg.qtables_from_quality(quality)[:4]), n_streams)
    n_mb = enc.total_blocks
    d_frames = ctx.alloc(n_frames * n_streams * fb)
    d_coef, d_mv, d_has = ctx.alloc(n_streams * n_mb * 512), ctx.alloc(n_streams * n_mb * 2), ctx.alloc(n_streams * n_mb)
    d_out = ctx.alloc(n_streams * fb)
    dec.set_output_dev(d_out)

    def fill(t0):
        for t in range(n_frames):
            ctx.synth_frames_dev(w, h, seeds, t0 + t, d_frames + t * n_streams * fb)

    fill(0)
    with pkg.Graph(ctx) as g:
        for t in range(n_frames):
            f = d_frames + t * n_streams * fb
            if t == 0:
                enc.encode_iframe_dev(f, d_coef)
                dec.decode_iframe_dev(d_coef)
            else:
                enc.encode_pframe_dev(f, d_mv, d_has, d_coef)
                dec.decode_pframe_dev(d_mv, d_has, d_coef)
    for rep, t0 in enumerate((0, 7, 3)):          # every replay on different content: stale results cannot pass
        fill(t0)
        g.launch()
        dec.check()
        recon, fbuf = enc.prev_frame(), dec.framebuffer()
        coef = np.empty((n_streams, n_mb, 256), np.int16)
        ctx.download(coef, d_coef)
        out = np.empty((n_streams, fb), np.uint8)
        ctx.download(out, d_out)
        for s, seed in enumerate(seeds):
            st = pkg.SyntheticStream(w, h, seed=seed)
            oenc = oracle.encoder(w, h, quality)
            oenc.encode_iframe(st.frame(t0))
            for t in range(1, n_frames):
                _, _, ocoef = oenc.encode_pframe(st.frame(t0 + t))
            assert np.array_equal(coef[s], ocoef), (rep, s)
            assert np.array_equal(recon[s], oenc.prev_frame()), (rep, s)
            assert np.array_equal(fbuf[s], recon[s]), (rep, s)
            pf = pkg.VideoFrame.from_packed(w, h, recon[s], padded=True)
            want = np.concatenate([pf.plane_y.image()[:h, :w].reshape(-1), pf.plane_u.image()[:h // 2, :w // 2].reshape(-1),
                                   pf.plane_v.image()[:h // 2, :w // 2].reshape(-1)])
            assert np.array_equal(out[s], want), (rep, s)
    g.close()
    enc.close()
    dec.close()
    for p in (d_frames, d_coef, d_mv, d_has, d_out):
        ctx.free(p)
