"""Stream-level checks shared by the GPU tests and the emulator tests: the product's
Encoder/Decoder (device hot path + host entropy/container) against the CPU oracle's, byte for byte."""
from __future__ import annotations

import io
import os

import numpy as np

from oracle_bind import OracleStreamDecoder, OracleStreamEncoder


def frame_of(pkg, w, h, packed):
    return pkg.VideoFrame.from_packed(w, h, packed)


def encode_clip(pkg, ctx, oracle, w, h, fps, quality, n_frames, gop, drop_at=(), frame_src=None, threads=1):
    """frame_src: t -> packed Y|U|V frame (default: synth.SyntheticStream); threads: the oracle's pool size"""
    if frame_src is None:
        frame_src = pkg.SyntheticStream(w, h).frame
    buf = io.BytesIO()
    enc = pkg.Encoder(buf, w, h, fps, quality, ctx)
    oenc = OracleStreamEncoder(oracle, w, h, fps, quality, threads=threads)
    for t in range(n_frames):
        f = frame_src(t)
        if t in drop_at:
            enc.encode_dropframe(); oenc.encode_dropframe()
        elif t % gop == 0:
            enc.encode_iframe(frame_of(pkg, w, h, f)); oenc.encode_iframe(f)
        else:
            enc.encode_pframe(frame_of(pkg, w, h, f)); oenc.encode_pframe(f)
    enc.finish(); oenc.finish()
    enc.close()
    return buf.getvalue(), oenc.bytes()


def check_stream_roundtrip(pkg, ctx, oracle, w, h, quality, n_frames, gop, drop_at=(), frame_src=None, threads=1):
    data, odata = encode_clip(pkg, ctx, oracle, w, h, 30, quality, n_frames, gop, drop_at, frame_src, threads)
    assert data[:8] == b"PFVIDEO\x00" and int.from_bytes(data[8:12], "little") == 211      # common.rs:1-2
    assert data[-5:] == b"\x00\x00\x00\x00\x00"                                              # EOF packet (enc.rs:221-227)
    assert data == odata, "product .pfv stream differs from the oracle's"
    # decode with the product and with the oracle: same frames, same count, both hit EOF; the product with the run streams read by the
    # host parser and by the device stage (every packet: PFV_ENTROPY_DECODE_DEVICE), which must hand over the same frames
    hdec = pkg.Decoder(io.BytesIO(data), ctx, entropy="host")
    hframes = []
    while hdec.advance_frame(lambda fr: hframes.append(fr.packed())):
        pass
    assert hdec.entropy_counts() == {"packets_read_on_device": 0, "packets_left_to_host_parser": 0}
    hdec.close()
    dec = pkg.Decoder(io.BytesIO(data), ctx, entropy="device")
    assert (dec.width(), dec.height(), dec.framerate()) == (w, h, 30)
    odec = OracleStreamDecoder(oracle, data, threads=threads)
    frames = []
    n_calls = 0
    while True:
        more = dec.advance_frame(lambda fr: frames.append(fr.packed()))
        n_calls += 1
        if not more:
            break
    counts = dec.entropy_counts()
    assert counts["packets_read_on_device"] + counts["packets_left_to_host_parser"] >= 1 or n_frames == len(drop_at), counts
    assert len(hframes) == len(frames) and all(np.array_equal(a, b) for a, b in zip(hframes, frames)), "host-parsed and device-read frames differ"
    oframes = []
    while True:
        rc, fr = odec.advance_frame()
        assert rc >= 0
        if fr is not None:
            oframes.append(fr)
        if rc == 0:
            break
    assert n_calls == n_frames + 1                                   # one call per packet, the last one sees EOF
    assert len(frames) == len(oframes) == n_frames - len(drop_at)    # drop frames produce no callback (dec.rs:190)
    for a, b in zip(frames, oframes):
        assert np.array_equal(a, b)
    assert dec.advance_frame(lambda fr: None) is False               # stays at EOF (dec.rs:171-173)
    # frames left in device memory (pfv_decoder_set_output_device): fetched from the address the callback gets
    ddec = pkg.Decoder(io.BytesIO(data), ctx)
    ddec.set_output_device(True)
    dframes = []

    def on_dev(addr):
        a = np.empty(w * h + 2 * (w // 2) * (h // 2), np.uint8)
        ctx.download(a, addr)
        dframes.append(a)
    while ddec.advance_frame(on_dev):
        pass
    ddec.close()
    assert len(dframes) == len(frames) and all(np.array_equal(a, b) for a, b in zip(dframes, frames)), "frames left in device memory differ"
    dec.reset()                                                      # dec.rs:148-152
    again = []
    assert dec.advance_frame(lambda fr: again.append(fr.packed())) is True
    assert np.array_equal(again[0], frames[0])
    dec.close()
    return data


def check_advance_delta(pkg, ctx, oracle, data, kinds, fps=30):
    """kinds: one entry per packet in stream order, True = a decodable frame, False = a drop frame.
    advance_delta(delta) consumes floor(accumulated / frame period) packets (dec.rs:154-167); drop-frame packets
    consume a period but produce no callback (dec.rs:190)."""
    dec = pkg.Decoder(data, ctx)
    got = []
    assert dec.advance_delta(2.5 / fps, lambda fr: got.append(1)) is True       # 2 periods now, half a period carried over
    assert len(got) == sum(kinds[:2])
    assert dec.advance_delta(0.6 / fps, lambda fr: got.append(1)) is True       # 0.5 + 0.6 -> one more packet
    assert len(got) == sum(kinds[:3])
    assert dec.advance_delta(100.0, lambda fr: got.append(1)) is False          # runs into EOF
    assert len(got) == sum(kinds)
    dec.close()


def check_header_errors(pkg, ctx, data):
    import pytest
    bad = bytearray(data); bad[0] = ord("Q")
    with pytest.raises(pkg.DecodeError) as e:
        pkg.Decoder(bytes(bad), ctx)
    assert e.value.code == pkg._lib.PFV_ERR_FORMAT
    bad = bytearray(data); bad[8] = 210
    with pytest.raises(pkg.DecodeError) as e:
        pkg.Decoder(bytes(bad), ctx)
    assert e.value.code == pkg._lib.PFV_ERR_VERSION
    with pytest.raises(pkg.DecodeError) as e:
        pkg.Decoder(data[:15], ctx)
    assert e.value.code == pkg._lib.PFV_ERR_IO
    # truncated payload -> I/O error while advancing, not a crash
    dec = pkg.Decoder(data[:len(data) // 2], ctx)
    with pytest.raises(pkg.PfvError) as e:
        while dec.advance_frame(lambda fr: None):
            pass
    assert e.value.code in (pkg._lib.PFV_ERR_IO, pkg._lib.PFV_ERR_FORMAT)
    dec.close()
    # unknown packet types are skipped (dec.rs:216-219)
    hdr = 20 + 4 * 128
    spliced = data[:hdr] + bytes([9]) + (3).to_bytes(4, "little") + b"abc" + data[hdr:]
    dec = pkg.Decoder(spliced, ctx)
    n = 0
    while dec.advance_frame(lambda fr: None):
        n += 1
    assert n >= 1
    dec.close()


def _outcomes_product(pkg, ctx, data, lookahead=None, entropy=None):
    """one entry per advance_frame call: ('frame', bytes) / ('none',) / ('eof',) / ('err', code)"""
    out = []
    try:
        dec = pkg.Decoder(data, ctx, lookahead=lookahead, entropy=entropy)
    except pkg.PfvError as e:
        return [("open-err", e.code)]
    try:
        for _ in range(64):
            got = []
            try:
                more = dec.advance_frame(lambda fr: got.append(fr.packed()))
            except pkg.PfvError as e:
                out.append(("err", e.code))
                break
            out.append(("frame", got[0].tobytes()) if got else ("none",))
            if not more:
                out.append(("eof",))
                break
    finally:
        dec.close()
    return out


def _outcomes_oracle(oracle, data):
    odec = OracleStreamDecoder(oracle, data)
    if not odec.h:
        return [("open-err", odec.err)]
    out = []
    for _ in range(64):
        rc, fr = odec.advance_frame()
        if rc < 0:
            out.append(("err", rc))
            break
        out.append(("frame", fr.tobytes()) if fr is not None else ("none",))
        if rc == 0:
            out.append(("eof",))
            break
    return out


def check_corrupted_streams(pkg, ctx, oracle, data, n_trials, seed):
    """Byte-flip fuzz of a valid .pfv stream: the product decoder and the oracle's must agree call by call -- the
    same frames (hostile coefficients and all), the same error code on the same packet, never a crash."""
    rng = np.random.default_rng(seed)
    hdr = 20 + 4 * 128
    stats = {"trials": 0, "errors": 0, "frames": 0}
    for _ in range(n_trials):
        bad = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(hdr, len(bad)))
            bad[pos] = int(rng.integers(0, 256))
        if rng.random() < 0.25:
            bad = bad[: int(rng.integers(hdr, len(bad)))]
        a = _outcomes_product(pkg, ctx, bytes(bad), lookahead=(None, 0, 3)[stats["trials"] % 3],   # default / inline / 3 workers
                              entropy=("device", "host")[(stats["trials"] // 3) % 2])                  # run streams read by the device stage / the host parser
        b = _outcomes_oracle(oracle, bytes(bad))
        assert len(a) == len(b), (a[-1][:1], b[-1][:1], [x[0] for x in a], [x[0] for x in b])
        for k, (x, y) in enumerate(zip(a, b)):
            assert x[0] == y[0], (k, x[0], y[0], x[1:] if x[0] == "err" else None, y[1:] if y[0] == "err" else None)
            if x[0] == "err":
                assert x[1] == y[1], (k, x, y)
            elif x[0] == "frame":
                assert x[1] == y[1], f"call {k}: decoded frames differ on a corrupted stream"
                stats["frames"] += 1
        stats["trials"] += 1
        stats["errors"] += a[-1][0] == "err"
    return stats


def check_lookahead_reset(pkg, ctx, data, n_frames):
    """reset() and set_lookahead in mid-stream drop whatever was parsed ahead; decoding restarts cleanly"""
    ref = [x[1] for x in _outcomes_product(pkg, ctx, data, lookahead=0) if x[0] == "frame"]
    assert len(ref) == n_frames
    dec = pkg.Decoder(data, ctx, lookahead=3)
    got = []
    for _ in range(2):
        assert dec.advance_frame(lambda fr: got.append(fr.packed().tobytes()))
    dec.reset()
    ctx.check(ctx._lib.pfv_decoder_set_lookahead(dec.handle, 1))
    assert dec.advance_frame(lambda fr: got.append(fr.packed().tobytes()))
    ctx.check(ctx._lib.pfv_decoder_set_lookahead(dec.handle, 2))      # mid-stream: position is kept
    while dec.advance_frame(lambda fr: got.append(fr.packed().tobytes())):
        pass
    assert got == ref[:2] + ref
    dec.close()


def check_batch_encoder(pkg, ctx, oracle, w, h, quality, n_streams, n_frames, gop):
    """BatchEncoder: every writer receives exactly the bytes the oracle's Encoder produces for that stream"""
    bufs = [io.BytesIO() for _ in range(n_streams)]
    enc = pkg.BatchEncoder(bufs, w, h, 30, quality, ctx)
    streams = [pkg.SyntheticStream(w, h, seed=pkg.synth.SEED + 5 * s) for s in range(n_streams)]
    oencs = [OracleStreamEncoder(oracle, w, h, 30, quality) for _ in range(n_streams)]
    for t in range(n_frames):
        for s, st in enumerate(streams):
            f = st.frame(t)
            enc.frames[s] = f
            (oencs[s].encode_iframe if t % gop == 0 else oencs[s].encode_pframe)(f)
        (enc.encode_iframes if t % gop == 0 else enc.encode_pframes)()
    enc.finish()
    enc.close()
    for s in range(n_streams):
        oencs[s].finish()
        assert bufs[s].getvalue() == oencs[s].bytes(), f"stream {s}: batch encoder bytes differ from the oracle's stream"


def check_batch_encoder_writer_failure(pkg, ctx, w=64, h=48, n_streams=3):
    """a writer that raises (disk full, closed pipe): the error reaches the caller of encode / flush / finish (the reference
    propagates every write error with `?`, src/enc.rs:190-235) instead of vanishing inside the ctypes callback; after the
    failure nothing more is written to ANY writer, so no stream has bytes behind a hole"""
    class Failing:
        def __init__(self, fail_after):
            self.buf, self.calls, self.fail_after = io.BytesIO(), 0, fail_after

        def write(self, b):
            self.calls += 1
            if self.calls > self.fail_after:
                raise OSError(28, "No space left on device")
            self.buf.write(b)

    st = pkg.SyntheticStream(w, h)
    for fail_after in (0, 1, 2):                      # the header write, the first packet, the second packet
        writers = [io.BytesIO(), Failing(fail_after), io.BytesIO()][:n_streams]
        raised = None
        try:
            enc = pkg.BatchEncoder(writers, w, h, 30, 5, ctx)
            for t in range(4):
                for s in range(n_streams):
                    enc.frames[s] = st.frame(t)
                (enc.encode_iframes if t == 0 else enc.encode_pframes)()
            enc.finish()
        except OSError as e:
            raised = e
        assert raised is not None and raised.errno == 28, f"writer failure after {fail_after} writes was swallowed"
        sizes = [len(wr.getvalue()) if hasattr(wr, "getvalue") else len(wr.buf.getvalue()) for wr in writers]
        enc.close()                                   # must not raise, must not write behind the hole
        assert sizes == [len(wr.getvalue()) if hasattr(wr, "getvalue") else len(wr.buf.getvalue()) for wr in writers]


def check_batch_decoder(pkg, ctx, oracle, w, h, quality, n_streams, n_frames, gop, noise=False):
    """BatchDecoder over the streams a BatchEncoder wrote: every step's frames equal the oracle decoder's, stream by stream.
    noise=True feeds white noise (dense coefficients: the decoder's sparse lists overflow and it parses the dense form)."""
    bufs = [io.BytesIO() for _ in range(n_streams)]
    enc = pkg.BatchEncoder(bufs, w, h, 30, quality, ctx)
    streams = [pkg.SyntheticStream(w, h, seed=pkg.synth.SEED + 3 * s) for s in range(n_streams)]
    rng = np.random.default_rng(w * h + quality)
    for t in range(n_frames):
        for s, st in enumerate(streams):
            enc.frames[s] = rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8) if noise else st.frame(t)
        (enc.encode_iframes if t % gop == 0 else enc.encode_pframes)()
    enc.finish(); enc.close()
    data = [b.getvalue() for b in bufs]
    for mode in ("host", "device"):       # the run streams read by the host pool / by the device's entropy stage (every step)
        odecs = [OracleStreamDecoder(oracle, d) for d in data]
        dec = pkg.BatchDecoder(data, ctx, threads=3, entropy=mode)
        assert (dec.width, dec.height, dec.framerate) == (w, h, 30)
        steps = 0
        while True:
            fr = dec.advance_frames()
            if fr is False:
                break
            for s in range(n_streams):
                rc, want = odecs[s].advance_frame()
                assert rc == 1 and np.array_equal(fr[s], want), f"step {steps} stream {s} (payloads read on the {mode})"
            steps += 1
        assert steps == n_frames and dec.advance_frames() is False
        for od in odecs:
            assert od.advance_frame()[0] == 0            # the oracle is at EOF too
        counts = dec.entropy_counts()
        if mode == "host":
            assert not noise or dec.dense_steps > 0      # white noise must have taken the dense path (other content may, too)
            assert counts == {"packets_read_on_device": 0, "packets_left_to_host_parser": 0}
        else:
            assert counts["packets_read_on_device"] + counts["packets_left_to_host_parser"] == n_streams * n_frames, counts
            assert counts["packets_read_on_device"] > 0, counts
        dec.close()


def check_encoder_keeps_nothing(pkg, ctx, w=48, h=32):
    """the native encoder must not accumulate the stream.  After every call the writer has received the packet
    and the library holds zero pending bytes (the reference writes each packet through, src/enc.rs:190-235)."""
    import ctypes
    st = pkg.SyntheticStream(w, h)
    buf = io.BytesIO()
    enc = pkg.Encoder(buf, w, h, 30, 5, ctx)
    sizes = [len(buf.getvalue())]
    assert sizes[0] == 20 + 4 * 128
    for t in range(4):
        (enc.encode_iframe if t == 0 else enc.encode_pframe)(frame_of(pkg, w, h, st.frame(t)))
        data, n = ctypes.c_void_p(), ctypes.c_size_t()
        ctx.check(ctx._lib.pfv_encoder_bytes(enc.handle, ctypes.byref(data), ctypes.byref(n)))
        assert n.value == 0                                  # nothing pending inside the library
        sizes.append(len(buf.getvalue()))
        assert sizes[-1] > sizes[-2]
    enc.finish()
    enc.close()
    assert buf.getvalue()[-5:] == bytes(5)


def check_empty_pframe_packet(pkg, ctx, oracle, w=48, h=32):
    """a type-2 packet with length 0 is not a drop frame (only type 1 is, src/dec.rs:188-202); the reference fails
    reading its payload.  Single-stream decoder, batch decoder and the oracle's decoder must agree that it is an error."""
    import pytest
    data, _ = encode_clip(pkg, ctx, oracle, w, h, 30, 5, n_frames=2, gop=15)
    hdr = 20 + 4 * 128
    n0 = int.from_bytes(data[hdr + 1:hdr + 5], "little")
    bad = data[:hdr + 5 + n0] + bytes([2, 0, 0, 0, 0]) + bytes(5)       # i-frame, empty p-frame packet, EOF
    odec = OracleStreamDecoder(oracle, bad)
    assert odec.advance_frame()[0] == 1
    assert odec.advance_frame()[0] < 0
    dec = pkg.Decoder(bad, ctx)
    assert dec.advance_frame(lambda fr: None) is True
    with pytest.raises(pkg.PfvError) as e:
        dec.advance_frame(lambda fr: None)
    assert e.value.code in (pkg._lib.PFV_ERR_IO, pkg._lib.PFV_ERR_FORMAT)
    dec.close()
    bd = pkg.BatchDecoder([bad, bad], ctx, threads=1)
    assert bd.advance_frames() is not None
    with pytest.raises(pkg.DecodeError) as e:
        bd.advance_frames()
    assert e.value.code == pkg._lib.PFV_ERR_IO
    bd.close()


# ---------------------------------------------------------------------------------------------------------------- GOP-batched objects
# where the GOP-batched decoder read its packet payloads, summed over every decoder _outcomes closed (PFV_OPT_ENTROPY_DECODE)
ENTROPY_COUNTS = {"packets_read_on_device": 0, "packets_left_to_host_parser": 0}
GOP_ENTROPY_MODES = ("host", "device")


def _outcomes(make_decoder, pkg, n_calls=96, stop_at_error=True):
    """one entry per advance_frame call of any decoder object: ('frame', bytes) / ('none',) / ('eof',) / ('err', code)"""
    out = []
    try:
        dec = make_decoder()
    except pkg.PfvError as e:
        return [("open-err", e.code)]
    try:
        for _ in range(n_calls):
            got = []
            try:
                more = dec.advance_frame(lambda fr: got.append(fr.packed()))
            except pkg.PfvError as e:
                out.append(("err", e.code))
                if stop_at_error:
                    break
                continue
            out.append(("frame", got[0].tobytes()) if got else ("none",))
            if not more:
                out.append(("eof",))
                break
    finally:
        if hasattr(dec, "stats"):
            for k, v in dec.stats().items():
                if k in ENTROPY_COUNTS:
                    ENTROPY_COUNTS[k] += v
        dec.close()
    return out


def encode_pattern(pkg, ctx, oracle, w, h, quality, pattern, make_encoder, frame_src=None, threads=1, with_oracle=True, keep_open=False):
    """pattern: a string of 'I' / 'P' / 'D' (drop frame), one per packet.  Returns (product bytes, oracle bytes or None).  keep_open: the
    encoder is finished but not closed (the caller wants its statistics)"""
    if frame_src is None:
        frame_src = pkg.SyntheticStream(w, h).frame
    buf = io.BytesIO()
    enc = make_encoder(buf)
    oenc = OracleStreamEncoder(oracle, w, h, 30, quality, threads=threads) if with_oracle else None
    t = 0
    for c in pattern:
        if c == "D":
            enc.encode_dropframe()
            if oenc: oenc.encode_dropframe()
            continue
        f = frame_src(t)
        t += 1
        if c == "I":
            enc.encode_iframe(frame_of(pkg, w, h, f))
            if oenc: oenc.encode_iframe(f)
        else:
            enc.encode_pframe(frame_of(pkg, w, h, f))
            if oenc: oenc.encode_pframe(f)
    enc.finish()
    if oenc: oenc.finish()
    if not keep_open:
        enc.close()
    return buf.getvalue(), (oenc.bytes() if oenc else None)


def check_gop_objects(pkg, ctx, oracle, w, h, quality, pattern, shapes, frame_src=None, threads=1, dec_threads=2, alternate_modes=False):
    """pfv_gop_encoder / pfv_gop_decoder against the frame-by-frame objects and the oracle on one packet pattern:
    for every batch shape (max_gops, max_gop_frames) the .pfv bytes equal the serial product Encoder's and the oracle's, and the
    GOP-batched decoder delivers, call by call, what the serial Decoder and the oracle's decoder deliver."""
    serial, odata = encode_pattern(pkg, ctx, oracle, w, h, quality, pattern, lambda buf: pkg.Encoder(buf, w, h, 30, quality, ctx), frame_src, threads)
    assert serial == odata, "serial product stream differs from the oracle's"
    want = _outcomes_oracle(oracle, serial)
    assert [x[0] for x in want].count("frame") == sum(c != "D" for c in pattern)
    for shape_no, (max_gops, max_len) in enumerate(shapes):
        data, _ = encode_pattern(pkg, ctx, oracle, w, h, quality, pattern,
                                 lambda buf: pkg.GopEncoder(buf, w, h, 30, quality, ctx, max_gops=max_gops, max_gop_frames=max_len), frame_src, with_oracle=False)
        assert data == serial, f"GOP-batched encoder (max_gops {max_gops}, max_gop_frames {max_len}) wrote a different .pfv stream"
        for mode in (GOP_ENTROPY_MODES[(shape_no + 1) % 2:][:1] if alternate_modes else GOP_ENTROPY_MODES):   # alternate: the slow emulator runs each shape under one reader
            got = _outcomes(lambda: pkg.GopDecoder(serial, ctx, max_gops=max_gops, max_gop_frames=max_len, threads=dec_threads, entropy=mode), pkg)
            assert len(got) == len(want), (max_gops, max_len, mode, [x[0] for x in got], [x[0] for x in want])
            for k, (a, b) in enumerate(zip(got, want)):
                assert a == b, f"GOP-batched decoder (max_gops {max_gops}, max_gop_frames {max_len}, payloads read on the {mode}): call {k} gives {a[0]}, the oracle {b[0]}"
    # frames taken from device memory (pfv_gop_encoder_encode_*_dev): the same stream
    if True:
        max_gops, max_len = shapes[0]
        fbytes = w * h + 2 * (w // 2) * (h // 2)
        src = frame_src if frame_src is not None else pkg.SyntheticStream(w, h).frame
        dev = ctx.alloc(fbytes)
        dbuf = io.BytesIO()
        denc = pkg.GopEncoder(dbuf, w, h, 30, quality, ctx, max_gops=max_gops, max_gop_frames=max_len)
        t = 0
        for c in pattern:
            if c == "D":
                denc.encode_dropframe()
                continue
            ctx.upload(dev, np.ascontiguousarray(src(t)))
            (denc.encode_iframe_dev if c == "I" else denc.encode_pframe_dev)(dev)
            t += 1
        denc.finish()
        denc.close()
        ctx.free(dev)
        assert dbuf.getvalue() == serial, "GOP-batched encoder fed from device memory wrote a different .pfv stream"
        # ... and BY REFERENCE (pfv_gop_encoder_set_frames_by_reference): every frame of the clip stays resident where the caller put it, the
        # batches' kernels read it there; one misaligned frame in between falls back to the copy.  For every batch shape.
        n_coded = sum(1 for c in pattern if c != "D")
        stride = (fbytes + 15) // 16 * 16 + 16                                        # every frame at a 16-byte aligned address of its own
        clip = ctx.alloc(n_coded * stride + 64)
        for shape_no, (max_gops, max_len) in enumerate(list(shapes) + [shapes[0]]):
            tiny_arena = shape_no == len(shapes)            # once more with an arena every batch outgrows: made again FROM the caller's frames
            rbuf = io.BytesIO()
            if tiny_arena:
                os.environ["PFV_TEST_GOP_ARENA_BYTES"] = "64"
            try:
                renc = pkg.GopEncoder(rbuf, w, h, 30, quality, ctx, max_gops=max_gops, max_gop_frames=max_len)
            finally:
                os.environ.pop("PFV_TEST_GOP_ARENA_BYTES", None)
            renc.set_frames_by_reference(True)
            t = 0
            for c in pattern:
                if c == "D":
                    renc.encode_dropframe()
                    continue
                at = clip + t * stride + (8 if t == 2 else 0)                         # frame 2 at an address that is not 16-byte aligned: copied
                ctx.upload(at, np.ascontiguousarray(src(t)))
                (renc.encode_iframe_dev if c == "I" else renc.encode_pframe_dev)(at)
                t += 1
            renc.finish()
            assert renc.stats()["frames_by_reference"] == n_coded - (1 if n_coded > 2 else 0), renc.stats()      # all but the misaligned one
            assert not tiny_arena or renc.stats()["batches_redone"] >= 1 or len(serial) < 64 * (len(pattern) + 2) + 1100   # (a clip whose batches fit 64 bytes has nothing to redo)
            renc.close()
            assert rbuf.getvalue() == serial, f"GOP-batched encoder reading device frames by reference (max_gops {max_gops}, max_gop_frames {max_len}) wrote a different stream"
        ctx.free(clip)
    # frames left in device memory (pfv_gop_decoder_set_output_device): the same bytes, fetched from the addresses the callback gets
    max_gops, max_len = shapes[-1]
    fb = w * h + 2 * (w // 2) * (h // 2)
    dd = pkg.GopDecoder(serial, ctx, max_gops=max_gops, max_gop_frames=max_len, threads=dec_threads, raw=True, entropy="device", output="device")
    got_dev = []

    def on_dev(y, u, v):
        assert u == y + w * h and v == u + (w // 2) * (h // 2)
        a = np.empty(fb, np.uint8)
        ctx.download(a, y)
        got_dev.append(a.tobytes())
    try:
        while dd.advance_frame(on_dev):
            pass
    except pkg.PfvError:
        pass
    dd.close()
    want_frames = [x[1] for x in want if x[0] == "frame"]
    assert got_dev == want_frames[:len(got_dev)] and len(got_dev) == len(want_frames), "frames delivered in device memory differ"
    # reset() (dec.rs:148-152) in mid-batch and real-time pacing (advance_delta, dec.rs:154-167): as the frame-by-frame Decoder behaves
    if pattern[:1] == "I" and len(pattern) >= 3:
        max_gops, max_len = shapes[0]
        gd, sd = pkg.GopDecoder(serial, ctx, max_gops=max_gops, max_gop_frames=max_len, threads=1, entropy="device"), pkg.Decoder(serial, ctx, lookahead=0)
        ga, sa = [], []
        for dec, acc in ((gd, ga), (sd, sa)):
            assert (dec.width(), dec.height(), dec.framerate()) == (w, h, 30)
            dec.advance_frame(lambda fr: acc.append(fr.packed().tobytes()))
            dec.advance_frame(lambda fr: acc.append(fr.packed().tobytes()))
            dec.reset()
            acc.append(dec.advance_delta(2.5 / 30, lambda fr: acc.append(fr.packed().tobytes())))    # two packets, half a period carried over
            acc.append(dec.advance_delta(0.6 / 30, lambda fr: acc.append(fr.packed().tobytes())))    # one more
            acc.append(dec.advance_delta(100.0, lambda fr: acc.append(fr.packed().tobytes())))       # runs into the end of the stream
            dec.close()
        assert ga == sa, "GOP-batched decoder: reset / advance_delta deliver something else than the frame-by-frame Decoder"
    return serial


def check_gop_encoder_flush_and_errors(pkg, ctx, oracle, w=64, h=48):
    """flush() in mid-GOP (the run continues in the next batch from the carried reference frame), packets only appear when a batch
    completes, finish twice / encode after finish are state errors, a payload budget too small is PFV_ERR_NOMEM and sticks"""
    import pytest
    st = pkg.SyntheticStream(w, h)
    pattern = "IPPPPIPP"
    serial, _ = encode_pattern(pkg, ctx, oracle, w, h, 5, pattern, lambda buf: pkg.Encoder(buf, w, h, 30, 5, ctx), with_oracle=False)
    buf = io.BytesIO()
    enc = pkg.GopEncoder(buf, w, h, 30, 5, ctx, max_gops=4, max_gop_frames=15)
    head = len(buf.getvalue())
    for t, c in enumerate(pattern):
        (enc.encode_iframe if c == "I" else enc.encode_pframe)(frame_of(pkg, w, h, st.frame(t)))
        if t == 2:
            assert len(buf.getvalue()) == head, "a packet left before its batch was complete"
            enc.flush()                                   # mid-GOP
            assert len(buf.getvalue()) > head
    enc.finish()
    with pytest.raises(AssertionError):
        enc.finish()
    rc = ctx._lib.pfv_gop_encoder_finish(enc.handle)
    assert rc == pkg._lib.PFV_ERR_STATE
    enc.close()
    assert buf.getvalue() == serial, "flush in mid-GOP changed the stream"
    # a caller that never drains: the library keeps everything (segments are copied out before their landing zone is reused) and
    # pfv_gop_encoder_bytes hands the whole stream over at the end
    import ctypes
    from pretty_fast_video_amd.context import ptr
    L = ctx._lib
    hnd = ctypes.c_void_p()
    ctx.check(L.pfv_gop_encoder_create(ctx.handle, w, h, 30, 5, 1, 2, 0, ctypes.byref(hnd)))
    for t, c in enumerate(pattern):
        fr = frame_of(pkg, w, h, st.frame(t))
        fn = L.pfv_gop_encoder_encode_iframe if c == "I" else L.pfv_gop_encoder_encode_pframe
        ctx.check(fn(hnd, ptr(fr.plane_y.pixels), ptr(fr.plane_u.pixels), ptr(fr.plane_v.pixels)))
    ctx.check(L.pfv_gop_encoder_finish(hnd))
    data, n = ctypes.c_void_p(), ctypes.c_size_t()
    ctx.check(L.pfv_gop_encoder_bytes(hnd, ctypes.byref(data), ctypes.byref(n)))
    assert ctypes.string_at(data.value, n.value) == serial, "undrained GOP encoder lost or reordered bytes"
    assert L.pfv_gop_encoder_batches(hnd) >= 4
    L.pfv_gop_encoder_destroy(hnd)
    # a batch whose payloads outgrow its landing zone (a sixth of the raw bytes + 16 KiB, pfv_gop.hip): noise at the finest quantiser (half
    # the raw bytes): the zone grows while the steps' payloads come over, what has arrived moves along
    rng = np.random.default_rng(77)
    fb = w * h + 2 * (w // 2) * (h // 2)
    noise = [rng.integers(0, 256, fb, dtype=np.uint8) for _ in range(12)]
    npat = ("I" + "P" * 7) * (int((16 << 10) / (0.3 * fb)) // 8 + 1)
    src = lambda t: noise[t % len(noise)]
    serial_n, _ = encode_pattern(pkg, ctx, oracle, w, h, 10, npat, lambda buf: pkg.Encoder(buf, w, h, 30, 10, ctx), src, with_oracle=False)
    assert len(serial_n) > len(npat) * fb // 6 + (16 << 10) + 4096, ("the case no longer outgrows the landing zone", len(serial_n), len(npat) * fb)
    got_n, _ = encode_pattern(pkg, ctx, oracle, w, h, 10, npat,
                              lambda buf: pkg.GopEncoder(buf, w, h, 30, 10, ctx, max_gops=len(npat) // 8, max_gop_frames=8), src, with_oracle=False)
    assert got_n == serial_n, "GOP encoder: a batch that outgrew its landing zone wrote a different stream"
    # no budget given: a batch that outgrows its arena is encoded again frame by frame and the arena grows -- like Encoder::encode_pframe
    # (src/enc.rs:125-173) the object cannot fail for size.  Real content never outgrows the default (binary noise at quality 0, the densest
    # there is: 1.5 x the raw bytes against the default's 2 x), so PFV_TEST_GOP_ARENA_BYTES (tests only) shrinks the default: batches of
    # every kind go through the redo -- a run cut in the middle (slot 0 continues from the saved reference frame), drop frames, leading
    # p-frames, device frames by reference -- and write the serial encoder's bytes.
    dense = [(rng.integers(0, 2, fb, dtype=np.uint8) * 255) for _ in range(8)]
    dsrc = lambda t: dense[t % len(dense)]
    for dpat, q, shape in (("IPPPPPPP" * 2, 0, (2, 8)), ("PPIPPDPPPPIPP", 5, (2, 3)), ("IPPPPPPPPPPP", 3, (1, 4))):
        serial_d, _ = encode_pattern(pkg, ctx, oracle, w, h, q, dpat, lambda buf: pkg.Encoder(buf, w, h, 30, q, ctx), dsrc, with_oracle=False)
        os.environ["PFV_TEST_GOP_ARENA_BYTES"] = "4096"
        try:
            holder = []

            def make(buf):
                holder.append(pkg.GopEncoder(buf, w, h, 30, q, ctx, max_gops=shape[0], max_gop_frames=shape[1]))
                return holder[-1]
            got_d, _ = encode_pattern(pkg, ctx, oracle, w, h, q, dpat, make, dsrc, with_oracle=False, keep_open=True)
        finally:
            del os.environ["PFV_TEST_GOP_ARENA_BYTES"]
        st_d = holder[-1].stats()
        holder[-1].close()
        assert got_d == serial_d, f"GOP encoder: batches made again after outgrowing their arena wrote a different stream ({dpat}, quality {q}, batch {shape})"
        assert st_d["batches_redone"] >= 1, st_d
    assert len(serial_d) > 0
    # an explicit payload budget is kept as given: 64 bytes cannot hold an i-frame
    enc = pkg.GopEncoder(io.BytesIO(), w, h, 30, 5, ctx, max_gops=2, max_gop_frames=4, payload_budget=64)
    enc.encode_iframe(frame_of(pkg, w, h, st.frame(0)))
    with pytest.raises(pkg.PfvError) as e:
        enc.flush()
    assert e.value.code == pkg._lib.PFV_ERR_NOMEM
    with pytest.raises(pkg.PfvError) as e:
        enc.encode_iframe(frame_of(pkg, w, h, st.frame(1)))
    assert e.value.code == pkg._lib.PFV_ERR_STATE
    enc.finished = True
    enc.close()


def check_gop_decoder_corrupted(pkg, ctx, oracle, data, n_trials, seed, shapes=((3, 15), (2, 2), (8, 4))):
    """Byte-flip fuzz: the GOP-batched decoder agrees with the oracle's decoder call by call up to the first error (same frames, same
    error code on the same packet) and with the product's frame-by-frame Decoder on EVERY call, errors included and beyond them --
    a packet that fails leaves the framebuffer alone; the p-frames behind a failed i-frame decode against the previous run's last
    frame, exactly as in the sequential loop."""
    rng = np.random.default_rng(seed)
    hdr = 20 + 4 * 128
    stats = {"trials": 0, "errors": 0, "frames": 0, "frames_after_an_error": 0}
    for _ in range(n_trials):
        bad = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(hdr, len(bad)))
            bad[pos] = int(rng.integers(0, 256))
        if rng.random() < 0.25:
            bad = bad[: int(rng.integers(hdr, len(bad)))]
        bad = bytes(bad)
        max_gops, max_len = shapes[stats["trials"] % len(shapes)]
        mode = GOP_ENTROPY_MODES[(stats["trials"] // len(shapes)) % len(GOP_ENTROPY_MODES)] if stats["trials"] % 4 else "device"
        mk = lambda: pkg.GopDecoder(bad, ctx, max_gops=max_gops, max_gop_frames=max_len, threads=stats["trials"] % 3, entropy=mode)
        a = _outcomes(mk, pkg)
        b = _outcomes_oracle(oracle, bad)
        assert len(a) == len(b), ([x[0] for x in a], [x[0] for x in b])
        for k, (x, y) in enumerate(zip(a, b)):
            assert x == y, (k, x[0], y[0], x[1:] if x[0] == "err" else None, y[1:] if y[0] == "err" else None)
        # beyond the first error: against the product's sequential decoder (the truncated tail of a stream errors forever: cap the calls)
        a2 = _outcomes(mk, pkg, n_calls=40, stop_at_error=False)
        b2 = _outcomes(lambda: pkg.Decoder(bad, ctx, lookahead=0), pkg, n_calls=40, stop_at_error=False)
        assert len(a2) == len(b2), ([x[0] for x in a2], [x[0] for x in b2])
        seen_err = False
        for k, (x, y) in enumerate(zip(a2, b2)):
            assert x == y, (max_gops, max_len, mode, k, [v[0] for v in a2], [v[0] for v in b2])
            seen_err = seen_err or x[0] == "err"
            stats["frames_after_an_error"] += seen_err and x[0] == "frame"
        stats["trials"] += 1
        stats["errors"] += any(x[0] == "err" for x in a2)
        stats["frames"] += sum(x[0] == "frame" for x in a2)
    return stats


def check_gop_decoder_dense_iframe_failure(pkg, ctx, oracle, w=124, h=212, quality=1, require_hit=True, shapes=((8, 15), (2, 2), (1, 15)), fracs=(0.55, 0.7, 0.8, 0.9, 0.97)):
    """Found by tools/soak.py (round 4): at a fine quantiser an i-frame is denser than 1 non-zero in 4, its coefficient LIST overflows
    before the parser reaches a corrupted byte further on, and the GOP-batched decoder took the frame for good when it cut its chains --
    the p-frames behind it then decoded against the slot's stale framebuffer instead of the previous run's last frame.  P P I P with the
    i-frame damaged in its second half: every call of the GOP-batched decoder must match the sequential Decoder's."""
    data, _ = encode_pattern(pkg, ctx, oracle, w, h, quality, "PPIP", lambda buf: pkg.Encoder(buf, w, h, 30, quality, ctx), with_oracle=False)
    hdr = 20 + 4 * 128
    pos, pk = hdr, []
    while pos + 5 <= len(data):
        n = int.from_bytes(data[pos + 1:pos + 5], "little")
        pk.append((data[pos], pos, n))
        pos += 5 + n
    assert [t for t, _, _ in pk] == [2, 2, 1, 2, 0]
    _, p, n = pk[2]
    hit = 0
    for frac in fracs:
        bad = bytearray(data)
        at = p + 5 + int(n * frac)
        bad[at:at + 4] = bytes(4)                                      # a hole of zero bits: a long run the frame has no room for
        bad[at + 4] ^= 0xFF
        bad = bytes(bad)
        want = _outcomes(lambda: pkg.Decoder(bad, ctx, lookahead=0), pkg, n_calls=12, stop_at_error=False)
        if [x[0] for x in want][:4] != ["frame", "frame", "err", "frame"]:
            continue                                                   # this flip happened to leave the packet parseable
        hit += 1
        for shape in shapes:
            for mode in GOP_ENTROPY_MODES:
                got = _outcomes(lambda: pkg.GopDecoder(bad, ctx, max_gops=shape[0], max_gop_frames=shape[1], threads=1, entropy=mode), pkg, n_calls=12, stop_at_error=False)
                assert got == want, (frac, shape, mode, [x[0] for x in got], [x == y for x, y in zip(got, want)])
    assert hit >= 1 or not require_hit, "no flip produced the failing i-frame this case is about"
    return hit


def check_gop_device_entropy(pkg, ctx, oracle, w, h, quality=5, pattern="IPPPPIPPPP", min_device_share=1.0, expect_unsettled=True, only=None):
    """The decoder's entropy stage on the device (k_entd_*, PFV_OPT_ENTROPY_DECODE) on VALID streams: (a) the synthetic pan content -- every
    packet's payload is read on the device and every call matches the oracle's decoder; (b) the same stream with the payload cut into 64-bit
    subsequences and one single round of reading (PFV_OPT_ENTDEC_*): the starts have not
    settled, packets go to the host parser by the 'unsettled' road -- same frames; (c) flat frames: every macroblock codes the same runs, the bit stream is periodic (a
    wrong read phase can persist) -- same frames whichever side reads them; (d) frames of noise at a fine quantiser: long codes and 15-bit
    values (the pair table's slow path)."""
    out = {}
    rng = np.random.default_rng(w * 7 + h)
    fb = w * h + 2 * (w // 2) * (h // 2)
    st = pkg.SyntheticStream(w, h)
    flat = [np.full(fb, 16 + 40 * (t // 3), np.uint8) for t in range(len(pattern))]
    noise = [rng.integers(0, 256, fb, dtype=np.uint8) for _ in range(len(pattern))]
    # pan_seams: 32-bit subsequences -- a small frame's payload then spans several workgroups of the full read, and the seams between them
    # are repaired by k_entd_fix (a lane of 32 bits holds a run or two: a wrong start takes many lanes to meet the true one, hence the rounds)
    cases = (("pan", st.frame, quality, None), ("pan_sub64", st.frame, quality, (64, 1, 1)), ("pan_seams", st.frame, quality, (32, None, 1024)),
             ("flat", lambda t: flat[t], quality, None), ("noise", lambda t: noise[t], 0, None), ("noise_seams", lambda t: noise[t], 0, (32, None, 1024)))
    for name, src, q, shape in cases:
        if only and name not in only:
            continue
        data, _ = encode_pattern(pkg, ctx, oracle, w, h, q, pattern, lambda buf: pkg.Encoder(buf, w, h, 30, q, ctx), src, with_oracle=False)
        want = _outcomes_oracle(oracle, data)
        n_packets = sum(c != "D" for c in pattern)
        assert [x[0] for x in want].count("frame") == n_packets
        dec = pkg.GopDecoder(data, ctx, max_gops=3, max_gop_frames=8, threads=2, entropy="device", entropy_shape=shape)
        got = []
        while True:
            fr = []
            more = dec.advance_frame(lambda f: fr.append(f.packed().tobytes()))
            got.append(("frame", fr[0]) if fr else ("none",))
            if not more:
                got.append(("eof",))
                break
        stats = dec.stats()
        dec.close()
        assert got == want, f"{name}: the decoder with the payloads read on the device delivers other frames than the oracle's"
        assert stats["packets_read_on_device"] + stats["packets_left_to_host_parser"] == n_packets, (name, stats)
        assert stats["left_unsettled"] + stats["left_irregular"] <= stats["packets_left_to_host_parser"]
        out[name] = {k: stats[k] for k in ("packets_read_on_device", "packets_left_to_host_parser", "left_unsettled", "left_irregular")}
    assert "pan" not in out or out["pan"]["packets_read_on_device"] >= min_device_share * n_packets, out
    assert "pan_sub64" not in out or out["pan_sub64"]["left_unsettled"] >= 1 or not expect_unsettled, out
    assert "pan_seams" not in out or out["pan_seams"]["packets_read_on_device"] >= min_device_share * n_packets, out
    assert "noise_seams" not in out or out["noise_seams"]["packets_read_on_device"] >= 1, out
    return out


def check_device_block_headers(pkg, ctx, oracle, w, h, quality=5, pattern="IPPP", seed=3):
    """A p-frame's block headers read on the device (k_hdr_*): frames large enough for several header workgroups, content that mixes
    macroblocks with and without motion vectors / coefficients (static background, moving noisy rectangles: 2-bit and 16-bit headers side by
    side) and the pan content (a vector on nearly every macroblock) -- every frame pfv_gop_decoder delivers with the payloads read on the
    device against the oracle's decoder, and every p-frame packet must have stayed on the device."""
    out = {}
    for kind in ("low_motion", "pan"):
        st = pkg.SyntheticStream(w, h, seed=pkg.synth.SEED + seed, kind=kind) if kind != "pan" else pkg.SyntheticStream(w, h)
        data, _ = encode_pattern(pkg, ctx, oracle, w, h, quality, pattern, lambda buf: pkg.Encoder(buf, w, h, 30, quality, ctx), st.frame, with_oracle=False)
        want = _outcomes_oracle(oracle, data)
        dec = pkg.GopDecoder(data, ctx, max_gops=2, max_gop_frames=8, threads=2, entropy="device")
        got = []
        while True:
            fr = []
            more = dec.advance_frame(lambda f: fr.append(f.packed().tobytes()))
            got.append(("frame", fr[0]) if fr else ("none",))
            if not more:
                got.append(("eof",))
                break
        stats = dec.stats()
        dec.close()
        assert got == want, f"{kind}: frames differ from the oracle's decoder"
        out[kind] = stats["packets_read_on_device"]
        assert stats["packets_read_on_device"] == len(pattern), (kind, stats)
    return out


def check_one_symbol_table_lists(pkg, ctx, oracle, w=64, h=48, seed=5):
    """A packet only the HOST parser can read AND whose coefficient list outgrows the room its bits were given: a code table with ONE symbol
    (symbol 1: every run is one zero and a 1-bit value, no code bits at all -- a value per payload bit where the lists count on three bits or
    more).  Valid for the reference (src/huffman.rs: a one-leaf tree reads without consuming bits).  Through pfv_gop_decoder and
    pfv_batch_decoder with the device entropy stage forced: the device leaves the packet to the host parser (degenerate table), the list
    overflows its place in the pool, is parsed again with room for every coefficient and gets a buffer of its own; every frame against the
    oracle's decoder and against the product's host-parser path."""
    rng = np.random.default_rng(seed)
    tabs = pkg.qtables_from_quality(5)
    head = b"PFVIDEO\0" + (211).to_bytes(4, "little") + b"".join(int(v).to_bytes(2, "little") for v in (w, h, 30, 4))
    head += b"".join(np.asarray(tabs[k], dtype="<u2").tobytes() for k in range(4))
    tb = int(pkg._lib.load().pfv_total_blocks(w, h))
    table = bytes([0, 255] + [0] * 14)

    def ipacket():       # tb x 128 runs of (1 zero, 1-bit value): tb x 128 bits
        bits = rng.integers(0, 2, tb * 128, dtype=np.uint8)
        return table + bytes([0, 1, 1]) + np.packbits(bits, bitorder="little").tobytes()

    def ppacket():       # block headers: no vector, has_coeff on every second macroblock; then 128 runs per coded macroblock
        hdr = np.zeros(tb * 2, np.uint8)
        hdr[1::4] = 1
        n_coded = int(hdr[1::2].sum())
        bits = np.concatenate([hdr, rng.integers(0, 2, n_coded * 128, dtype=np.uint8)])
        return table + bytes([2, 3, 3]) + np.packbits(bits, bitorder="little").tobytes()
    packets = [(1, ipacket()), (2, ppacket()), (2, ppacket()), (1, ipacket()), (2, ppacket())]
    data = head + b"".join(bytes([t]) + len(p).to_bytes(4, "little") + p for t, p in packets) + bytes(5)
    want = _outcomes_oracle(oracle, data)
    assert [x[0] for x in want].count("frame") == len(packets), "the oracle's decoder must accept the stream"
    for mode in ("host", "device"):
        dec = pkg.GopDecoder(data, ctx, max_gops=2, max_gop_frames=4, threads=2, entropy=mode)
        got = []
        while True:
            fr = []
            more = dec.advance_frame(lambda f: fr.append(f.packed().tobytes()))
            got.append(("frame", fr[0]) if fr else ("none",))
            if not more:
                got.append(("eof",))
                break
        stats = dec.stats()
        dec.close()
        assert got == want, f"pfv_gop_decoder ({mode}): frames differ from the oracle's"
        if mode == "device":
            assert stats["packets_left_to_host_parser"] == len(packets) and stats["packets_read_on_device"] == 0, stats
            assert stats["lists_spilled"] == len(packets), stats          # every one of them outgrew its place
    bdec = pkg.BatchDecoder([data, data], ctx, threads=2, entropy="device")
    k = 0
    while True:
        fr = bdec.advance_frames()
        if fr is False:
            break
        while want[k][0] != "frame":
            k += 1
        assert fr[0].tobytes() == want[k][1] and fr[1].tobytes() == want[k][1], f"pfv_batch_decoder: step {k} differs from the oracle's"
        k += 1
    assert bdec.entropy_counts()["packets_left_to_host_parser"] == 2 * len(packets)
    bdec.close()
    return len(packets)
