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
