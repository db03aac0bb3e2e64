"""Encoder -- mirror of ``pfv_rs::enc::Encoder`` (src/enc.rs:12-188).

``Encoder(writer, width, height, framerate, quality, ctx)``: the reference's ``num_threads`` slot is the
:class:`Context` (device + stream).  ``writer`` is any object with ``write(bytes)``; like the reference the header
is written on construction, one packet per ``encode_*`` call, the EOF packet on ``finish()`` (also on close /
garbage collection if the caller did not finish, src/enc.rs:28-34).
"""
from __future__ import annotations

import ctypes

from .context import Context, ptr
from .frame import VideoFrame


class Encoder:
    def __init__(self, writer, width: int, height: int, framerate: int, quality: int, ctx: Context, device_entropy: bool = True):
        assert 0 <= quality <= 10                                   # src/enc.rs:38
        self.ctx, self.writer = ctx, writer
        self.width, self.height = int(width), int(height)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.pfv_encoder_create(ctx.handle, self.width, self.height, int(framerate), int(quality), ctypes.byref(h)))
        self.handle = h
        # packet payloads from the device entropy stage (default) or the host serialisers: same bytes
        ctx.check(ctx._lib.pfv_encoder_set_device_entropy(h, 1 if device_entropy else 0))
        self._flushed = 0
        self.finished = False
        ctx._sessions.add(self)
        self._flush()                                               # header (src/enc.rs:70)

    def _flush(self):
        data, n = ctypes.c_void_p(), ctypes.c_size_t()
        self.ctx.check(self.ctx._lib.pfv_encoder_bytes(self.handle, ctypes.byref(data), ctypes.byref(n)))
        if n.value > self._flushed:
            self.writer.write(ctypes.string_at(data.value + self._flushed, n.value - self._flushed))
            self._flushed = n.value

    def _check_frame(self, frame: VideoFrame):
        assert frame.width == self.width and frame.height == self.height                                  # src/enc.rs:76-79
        assert frame.plane_y.width == frame.width and frame.plane_y.height == frame.height
        assert frame.plane_u.width == frame.width // 2 and frame.plane_u.height == frame.height // 2
        assert frame.plane_v.width == frame.width // 2 and frame.plane_v.height == frame.height // 2
        assert not self.finished                                                                          # src/enc.rs:80

    def encode_iframe(self, frame: VideoFrame):
        self._check_frame(frame)
        self.ctx.check(self.ctx._lib.pfv_encoder_encode_iframe(self.handle, ptr(frame.plane_y.pixels), ptr(frame.plane_u.pixels),
                                                               ptr(frame.plane_v.pixels)))
        self._flush()

    def encode_pframe(self, frame: VideoFrame):
        self._check_frame(frame)
        self.ctx.check(self.ctx._lib.pfv_encoder_encode_pframe(self.handle, ptr(frame.plane_y.pixels), ptr(frame.plane_u.pixels),
                                                               ptr(frame.plane_v.pixels)))
        self._flush()

    def encode_dropframe(self):
        assert not self.finished
        self.ctx.check(self.ctx._lib.pfv_encoder_encode_dropframe(self.handle))
        self._flush()

    def finish(self):
        assert not self.finished                                    # src/enc.rs:183
        self.ctx.check(self.ctx._lib.pfv_encoder_finish(self.handle))
        self.finished = True
        self._flush()

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            if not self.finished:                                   # impl Drop (src/enc.rs:28-34)
                self.finish()
            self.ctx._lib.pfv_encoder_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
