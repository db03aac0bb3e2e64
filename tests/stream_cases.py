"""Stream-level checks shared by the GPU tests and the emulator tests: the product's
Encoder/Decoder (device hot path + host entropy/container) against the CPU oracle's, byte for byte."""
from __future__ import annotations

import io

import numpy as np

from oracle_bind import OracleStreamDecoder, OracleStreamEncoder


def frame_of(pkg, w, h, packed):
    return pkg.VideoFrame.from_packed(w, h, packed)


def encode_clip(pkg, ctx, oracle, w, h, fps, quality, n_frames, gop, drop_at=()):
    st = pkg.SyntheticStream(w, h)
    buf = io.BytesIO()
    enc = pkg.Encoder(buf, w, h, fps, quality, ctx)
    oenc = OracleStreamEncoder(oracle, w, h, fps, quality)
    for t in range(n_frames):
        f = st.frame(t)
        if t in drop_at:
            enc.encode_dropframe(); oenc.encode_dropframe()
        elif t % gop == 0:
            enc.encode_iframe(frame_of(pkg, w, h, f)); oenc.encode_iframe(f)
        else:
            enc.encode_pframe(frame_of(pkg, w, h, f)); oenc.encode_pframe(f)
    enc.finish(); oenc.finish()
    enc.close()
    return buf.getvalue(), oenc.bytes()


def check_stream_roundtrip(pkg, ctx, oracle, w, h, quality, n_frames, gop, drop_at=()):
    data, odata = encode_clip(pkg, ctx, oracle, w, h, 30, quality, n_frames, gop, drop_at)
    assert data[:8] == b"PFVIDEO\x00" and int.from_bytes(data[8:12], "little") == 211      # common.rs:1-2
    assert data[-5:] == b"\x00\x00\x00\x00\x00"                                              # EOF packet (enc.rs:221-227)
    assert data == odata, "product .pfv stream differs from the oracle's"
    # decode with the product and with the oracle: same frames, same count, both hit EOF
    dec = pkg.Decoder(io.BytesIO(data), ctx)
    assert (dec.width(), dec.height(), dec.framerate()) == (w, h, 30)
    odec = OracleStreamDecoder(oracle, data)
    frames = []
    n_calls = 0
    while True:
        more = dec.advance_frame(lambda fr: frames.append(fr.packed()))
        n_calls += 1
        if not more:
            break
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
