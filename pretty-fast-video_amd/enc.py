"""Encoder -- mirror of ``pfv_rs::enc::Encoder`` (src/enc.rs:12-188).

``Encoder(writer, width, height, framerate, quality, ctx)``: the reference's ``num_threads`` slot is the
:class:`Context` (device + stream).  ``writer`` is any object with ``write(bytes)``; like the reference the header
is written on construction, one packet per ``encode_*`` call, the EOF packet on ``finish()`` (also on close /
garbage collection if the caller did not finish, src/enc.rs:28-34).
"""
from __future__ import annotations

import ctypes

from . import _lib
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
        self.finished = False
        ctx._sessions.add(self)
        self._flush()                                               # header (src/enc.rs:70)

    def _flush(self):
        # the library keeps only what has not been handed over yet (the reference writes each packet through, src/enc.rs:190-235)
        data, n = ctypes.c_void_p(), ctypes.c_size_t()
        self.ctx.check(self.ctx._lib.pfv_encoder_drain(self.handle, ctypes.byref(data), ctypes.byref(n)))
        if n.value:
            self.writer.write(ctypes.string_at(data.value, n.value))

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


class BatchEncoder:
    """``n`` independent streams of one geometry encoded together: one upload, one kernel launch per stage and one
    download per frame step for all of them (the reference runs one ``Encoder`` per stream, src/enc.rs:12-26).  Each
    writer receives exactly the bytes an ``Encoder`` of its own would have written.

    The caller fills ``frames`` -- a page-locked ``[n, frame_bytes]`` uint8 array, one packed Y|U|V frame per stream --
    and calls ``encode_iframes()`` / ``encode_pframes()``; ``finish()`` writes the EOF packets."""

    def __init__(self, writers, width: int, height: int, framerate: int, quality: int, ctx: Context):
        import numpy as np
        from .session import EncoderSession, qtables_from_quality
        assert 0 <= quality <= 10 and len(writers) >= 1
        self.ctx, self.writers, self.n = ctx, list(writers), len(writers)
        self.width, self.height = int(width), int(height)
        self.session = EncoderSession(ctx, width, height, quality, self.n)
        self.session.enable_entropy()
        s = self.session
        self.frames = ctx.host_array(self.n * s.frame_bytes).reshape(self.n, s.frame_bytes)
        self._cap = cap = (int(ctx._lib.pfv_payload_worst_case(width, height)) + 15) & ~15
        self._payloads = ctx.host_array(min(self.n * cap, max(self.n * s.frame_bytes, 1 << 20)))
        self._d_frames = ctx.alloc(self.n * s.frame_bytes)
        self._d_coef, self._d_mv, self._d_has = (ctx.alloc(self.n * s.total_blocks * 512), ctx.alloc(self.n * s.total_blocks * 2),
                                                 ctx.alloc(self.n * s.total_blocks))
        self.finished = False
        # header (src/enc.rs:190-219): magic, version, geometry, the four q-tables
        q = qtables_from_quality(quality)
        head = (b"PFVIDEO\x00" + (211).to_bytes(4, "little") + self.width.to_bytes(2, "little") + self.height.to_bytes(2, "little")
                + int(framerate).to_bytes(2, "little") + (4).to_bytes(2, "little")
                + b"".join(np.asarray(t, dtype="<u2").tobytes() for t in q[:4]))
        for w in self.writers:
            w.write(head)

    def _step(self, pframe: bool, frames=None):
        assert not self.finished
        s, ctx = self.session, self.ctx
        src = self.frames if frames is None else frames      # any [n, frame_bytes] uint8 array; page-locked ones upload fastest
        assert src.size == self.n * s.frame_bytes
        ctx.upload(self._d_frames, src)
        if pframe:
            s.encode_pframe_dev(self._d_frames, self._d_mv, self._d_has, self._d_coef)
            s.pack_pframe_dev(self._d_mv, self._d_has, self._d_coef)
        else:
            s.encode_iframe_dev(self._d_frames, self._d_coef)
            s.pack_iframe_dev(self._d_coef)
        try:
            sizes, offsets = s.payloads(self._payloads)
        except _lib.PfvError as e:                                   # very dense content: retry with the worst-case buffer
            if e.code != _lib.PFV_ERR_NOMEM or self._payloads.size >= self.n * self._cap:
                raise
            self._payloads = ctx.host_array(self.n * self._cap)
            sizes, offsets = s.payloads(self._payloads)
        kind = bytes([2 if pframe else 1])
        for w, n, o in zip(self.writers, sizes.tolist(), offsets.tolist()):
            w.write(kind + int(n).to_bytes(4, "little"))             # packet header (src/enc.rs:301-305, :453-457)
            w.write(self._payloads[o:o + n].data)

    def encode_iframes(self, frames=None):
        self._step(False, frames)

    def encode_pframes(self, frames=None):
        self._step(True, frames)

    def finish(self):
        assert not self.finished
        self.finished = True
        for w in self.writers:
            w.write(bytes(5))                                        # EOF packet (src/enc.rs:221-227)

    def close(self):
        if getattr(self, "session", None) is not None:
            if not self.finished:
                self.finish()
            for p in (self._d_frames, self._d_coef, self._d_mv, self._d_has):
                self.ctx.free(p)
            self.session.close()
            self.session = None
