"""Decoder -- mirror of ``pfv_rs::dec::Decoder`` (src/dec.rs:15-224).

``Decoder(reader, ctx)``: ``reader`` is a bytes-like object or anything with ``read()``; the ``num_threads`` slot
is the :class:`Context`.  ``advance_frame(onvideo)`` / ``advance_delta(delta, onvideo)`` call ``onvideo(frame)`` with
a :class:`VideoFrame` for every decoded frame and return ``True`` while there is more data, ``False`` at EOF.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .context import Context
from .frame import VideoFrame

_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)


class DecodeError(_lib.PfvError):
    """FormatError / VersionError / IOError of src/dec.rs:30-35 (see .code)"""


class Decoder:
    def __init__(self, reader, ctx: Context, lookahead: int | None = None):
        data = reader.read() if hasattr(reader, "read") else bytes(reader)
        self._data = np.frombuffer(data, dtype=np.uint8).copy()     # must outlive the native decoder
        self.ctx = ctx
        h = ctypes.c_void_p()
        rc = ctx._lib.pfv_decoder_create(ctx.handle, self._data.ctypes.data_as(ctypes.c_void_p), self._data.size, ctypes.byref(h))
        if rc != _lib.PFV_OK:
            msg = ctx._lib.pfv_last_error(ctx.handle)
            raise DecodeError(rc, msg.decode() if msg else "")
        self.handle = h
        ctx._sessions.add(self)
        if lookahead is not None:                                   # packets parsed ahead on this many worker threads
            ctx.check(ctx._lib.pfv_decoder_set_lookahead(h, int(lookahead)))

    def width(self) -> int:
        return self.ctx._lib.pfv_decoder_width(self.handle)

    def height(self) -> int:
        return self.ctx._lib.pfv_decoder_height(self.handle)

    def framerate(self) -> int:
        return self.ctx._lib.pfv_decoder_framerate(self.handle)

    def reset(self):
        self.ctx.check(self.ctx._lib.pfv_decoder_reset(self.handle))

    def _callback(self, onvideo):
        def cb(_user, y, u, v, w, h):
            def arr(p, n):
                return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
            cw, ch = w // 2, h // 2
            from .plane import VideoPlane
            onvideo(VideoFrame(w, h, VideoPlane.from_slice(w, h, arr(y, w * h)), VideoPlane.from_slice(cw, ch, arr(u, cw * ch)),
                               VideoPlane.from_slice(cw, ch, arr(v, cw * ch))))
        return _CB(cb)

    def advance_frame(self, onvideo) -> bool:
        cb = self._callback(onvideo)
        rc = self.ctx._lib.pfv_decoder_advance_frame(self.handle, cb, None)
        if rc < 0:
            self.ctx.check(rc)
        return rc == 1

    def advance_delta(self, delta: float, onvideo) -> bool:
        cb = self._callback(onvideo)
        rc = self.ctx._lib.pfv_decoder_advance_delta(self.handle, float(delta), cb, None)
        if rc < 0:
            self.ctx.check(rc)
        return rc == 1

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.pfv_decoder_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchDecoder:
    """``n`` ``.pfv`` streams of one geometry and one frame-type pattern (e.g. the outputs of :class:`BatchEncoder`)
    decoded together: the packets of a step are bit-parsed on a thread pool (one task per stream, the C parser releases
    the GIL), their non-zero coefficients go up as ONE sparse list and ONE kernel launch decodes all streams.

    ``advance_frames()`` returns the ``[n, frame_bytes]`` array of decoded frames (a view into page-locked memory, valid
    until the next call), ``None`` for a step of drop frames, or ``False`` at the end of the streams."""

    HEADER = 20

    def __init__(self, streams, ctx: Context, threads: int = 8):
        from concurrent.futures import ThreadPoolExecutor
        from .session import DecoderSession
        self.ctx = ctx
        self.data = [np.frombuffer(bytes(s.read() if hasattr(s, "read") else s), dtype=np.uint8) for s in streams]
        self.n = len(self.data)
        heads = {d[:self.HEADER + 128 * int(d[18]) + 128 * 256 * int(d[19])].tobytes() for d in self.data}
        if len(heads) != 1:
            raise ValueError("BatchDecoder: the streams must share one header (geometry, frame rate, q-tables)")
        d0 = self.data[0]
        if d0[:8].tobytes() != b"PFVIDEO\x00" or int.from_bytes(d0[8:12].tobytes(), "little") != 211:
            raise DecodeError(_lib.PFV_ERR_FORMAT, "bad magic or version (src/dec.rs:50-59)")
        u16 = lambda o: int(d0[o]) | int(d0[o + 1]) << 8
        self.width, self.height, self.framerate, self.n_qtables = u16(12), u16(14), u16(16), u16(18)
        q = np.frombuffer(d0[20:20 + 128 * self.n_qtables].tobytes(), dtype="<u2").astype(np.int32).reshape(-1, 64)
        self.session = DecoderSession(ctx, self.width, self.height, q, self.n)
        s = self.session
        self.pos = [self.HEADER + 128 * self.n_qtables] * self.n
        tb = s.total_blocks
        self._cap = tb * 256 // 4                                    # per stream: denser than 1 in 4 -> dense fallback
        self._idx = ctx.host_array(self.n * self._cap * 4).view(np.uint32)
        self._val = ctx.host_array(self.n * self._cap * 2).view(np.int16)
        self._mv = ctx.host_array(self.n * tb * 2).view(np.int8).reshape(self.n, tb, 2)
        self._has = ctx.host_array(self.n * tb).reshape(self.n, tb)
        self._frames = ctx.host_array(self.n * s.frame_bytes).reshape(self.n, s.frame_bytes)
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads)))
        self.eof = False

    def _next_packet(self, k):
        """(type, payload view) of stream k's next frame packet; unknown packet types are skipped (src/dec.rs:216-219)"""
        d = self.data[k]
        while True:
            p = self.pos[k]
            if p + 5 > d.size:
                raise DecodeError(_lib.PFV_ERR_IO, "unexpected end of stream in a packet header")
            typ, n = int(d[p]), int.from_bytes(d[p + 1:p + 5].tobytes(), "little")
            if typ == 0:
                return 0, None
            if p + 5 + n > d.size:
                raise DecodeError(_lib.PFV_ERR_IO, "packet payload runs past the end of the stream")
            self.pos[k] = p + 5 + n
            if typ in (1, 2):
                return typ, d[p + 5:p + 5 + n]

    def _parse(self, k, typ, payload):
        lib, tb = self.ctx._lib, self.session.total_blocks
        n, qidx = ctypes.c_size_t(), np.zeros(3, np.uint8)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        idx, val = self._idx[k * self._cap:(k + 1) * self._cap], self._val[k * self._cap:(k + 1) * self._cap]
        rc = lib.pfv_parse_payload_sparse(int(typ == 2), P(payload), payload.size, tb, self.n_qtables, P(self._mv[k]), P(self._has[k]),
                                          P(idx), P(val), self._cap, ctypes.byref(n), P(qidx))
        if rc == 1:
            return None, qidx.tobytes()                               # denser than 1 in 4: this step goes the dense way
        if rc:
            raise DecodeError(rc, "malformed packet payload")
        idx[:n.value] += np.uint32(k * tb * 256)                      # flat index into [stream][macroblock][256]
        return n.value, qidx.tobytes()

    def advance_frames(self):
        if self.eof:
            return False
        pk = [self._next_packet(k) for k in range(self.n)]
        types = {t for t, _ in pk}
        if len(types) != 1:
            raise ValueError("BatchDecoder: the streams' packet types diverge at this step")
        typ = types.pop()
        if typ == 0:
            self.eof = True
            return False
        if typ == 1 and all(p.size == 0 for _, p in pk):              # drop frames: an i-frame packet without payload (src/dec.rs:188-202)
            return None
        if typ == 2 and any(p.size == 0 for _, p in pk):              # an empty p-frame packet is a truncated read in the reference (src/dec.rs:204-214)
            raise DecodeError(_lib.PFV_ERR_IO, "p-frame packet without payload")
        if any(p.size == 0 for _, p in pk):
            raise ValueError("BatchDecoder: drop frames must line up across the streams")
        res = list(self._pool.map(lambda k: self._parse(k, typ, pk[k][1]), range(self.n)))
        if len({q for _, q in res}) != 1:
            raise ValueError("BatchDecoder: the streams use different q-table indices in this step")
        if any(c is None for c, _ in res):
            return self._dense_step(typ, pk, np.frombuffer(res[0][1], np.uint8))
        # compact the per-stream lists into one (they are ascending within and across streams)
        counts = [c for c, _ in res]
        at = 0
        for k, c in enumerate(counts):
            if at != k * self._cap:
                self._idx[at:at + c] = self._idx[k * self._cap:k * self._cap + c]
                self._val[at:at + c] = self._val[k * self._cap:k * self._cap + c]
            at += c
        qidx = np.frombuffer(res[0][1], np.uint8)
        s = self.session
        if typ == 1:
            s.decode_iframe_sparse(self._idx[:at], self._val[:at], qidx)
        else:
            s.decode_pframe_sparse(self._mv, self._has, self._idx[:at], self._val[:at], qidx)
        self.ctx.check(self.ctx._lib.pfv_dec_get_frame(s.handle, self._frames.ctypes.data_as(ctypes.c_void_p)))
        return self._frames

    def _dense_step(self, typ, pk, qidx):
        """some packet overflowed its sparse list: parse every stream into the dense [macroblock][256] form instead"""
        lib, s, tb = self.ctx._lib, self.session, self.session.total_blocks
        if getattr(self, "_coef", None) is None:
            self._coef = self.ctx.host_array(self.n * tb * 512).view(np.int16).reshape(self.n, tb, 256)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

        def parse(k):
            q = np.zeros(3, np.uint8)
            payload = pk[k][1]
            rc = (lib.pfv_parse_pframe_payload(P(payload), payload.size, tb, self.n_qtables, P(self._mv[k]), P(self._has[k]),
                                               P(self._coef[k]), P(q)) if typ == 2 else
                  lib.pfv_parse_iframe_payload(P(payload), payload.size, tb, self.n_qtables, P(self._coef[k]), P(q)))
            if rc:
                raise DecodeError(rc, "malformed packet payload")
        list(self._pool.map(parse, range(self.n)))
        if typ == 1:
            s.decode_iframe(self._coef, qidx)
        else:
            s.decode_pframe(self._mv, self._has, self._coef, qidx)
        self.ctx.check(lib.pfv_dec_get_frame(s.handle, self._frames.ctypes.data_as(ctypes.c_void_p)))
        return self._frames

    def close(self):
        if getattr(self, "session", None) is not None:
            self._pool.shutdown()
            self.session.close()
            self.session = None
