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
    writer receives exactly the bytes an ``Encoder`` of its own would have written.  The orchestration (copy stream for
    the next step's upload, payload collection, packet assembly) is the C++ ``pfv_batch_encoder``; this class marshals.

    The caller fills ``frames`` -- a page-locked ``[n, frame_bytes]`` uint8 array, one packed Y|U|V frame per stream; two
    such arrays alternate, fetch the attribute again after every step -- and calls ``encode_iframes()`` /
    ``encode_pframes()``; or passes its own ``[n, frame_bytes]`` array to them.  Packets reach the writers one step late;
    ``finish()`` flushes and writes the EOF packets."""

    def __init__(self, writers, width: int, height: int, framerate: int, quality: int, ctx: Context):
        import numpy as np
        assert 0 <= quality <= 10 and len(writers) >= 1
        self.ctx, self.writers, self.n = ctx, list(writers), len(writers)
        self.width, self.height = int(width), int(height)
        self.frame_bytes = int(ctx._lib.pfv_frame_bytes(width, height))
        self.finished = False
        self._np = np

        self._write_error = None          # ctypes swallows exceptions raised inside a callback: keep the first one, stop
        #                                   writing (nothing after a failed write may reach any writer out of order) and re-raise it
        #                                   when the C call returns -- the reference propagates every write error (`?`, src/enc.rs:190-235)

        def on_write(_user, stream, data, length):
            if self._write_error is not None:
                return
            try:
                self.writers[stream].write(ctypes.string_at(data, length))
            except BaseException as e:      # noqa: BLE001 -- re-raised by _raise_write_error
                self._write_error = e
        self._cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t)(on_write)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.pfv_batch_encoder_create(ctx.handle, self.width, self.height, int(framerate), int(quality), self.n,
                                                    ctypes.cast(self._cb, ctypes.c_void_p), None, ctypes.byref(h)))
        self.handle = h
        ctx._sessions.add(self)

    @property
    def frames(self):
        """the page-locked [n, frame_bytes] array to fill for the next step"""
        p = self.ctx._lib.pfv_batch_encoder_frames(self.handle)
        return self._np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(self.n, self.frame_bytes))

    def _step(self, pframe: bool, frames=None):
        assert not self.finished
        src = None
        if frames is not None:
            src = self._np.ascontiguousarray(frames, dtype=self._np.uint8)
            assert src.size == self.n * self.frame_bytes
        rc = self.ctx._lib.pfv_batch_encoder_encode(self.handle, 1 if pframe else 0, ptr(src) if src is not None else None)
        self._raise_write_error()
        self.ctx.check(rc)

    def _raise_write_error(self):
        """a writer failed inside the callback: the stream it belongs to is truncated, nothing more is written to any writer"""
        if self._write_error is not None:
            self.finished = True            # close() must not try to write EOF packets behind the hole
            raise self._write_error

    def encode_iframes(self, frames=None):
        self._step(False, frames)

    def encode_pframes(self, frames=None):
        self._step(True, frames)

    def flush(self):
        rc = self.ctx._lib.pfv_batch_encoder_flush(self.handle)
        self._raise_write_error()
        self.ctx.check(rc)

    def finish(self):
        assert not self.finished
        rc = self.ctx._lib.pfv_batch_encoder_finish(self.handle)
        self._raise_write_error()
        self.ctx.check(rc)
        self.finished = True

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            if not self.finished:
                self.finish()
            self.ctx._lib.pfv_batch_encoder_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _IoVec(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_size_t)]


class GopEncoder(Encoder):
    """:class:`Encoder` for one stream with the independent GOPs of the stream as the slots of every kernel launch
    (``pfv_gop_encoder``, include/pfv_hip.h): same calls, same ``.pfv`` bytes; a packet reaches the writer when its batch of
    ``max_gops`` groups is complete (or on ``flush()`` / ``finish()``).  ``encode_*frame`` also accept the three planes as flat
    uint8 arrays ``(y, u, v)`` -- e.g. views into page-locked memory (``Context.host_array``), which upload at PCIe rate."""

    def __init__(self, writer, width: int, height: int, framerate: int, quality: int, ctx: Context, max_gops: int = 8, max_gop_frames: int = 15,
                 payload_budget: int = 0, zero_copy: bool = False):
        """zero_copy: the writer receives memoryviews into the library's page-locked buffers (valid during the write() call only, which
        is all a file or a socket needs) instead of bytes copies"""
        assert 0 <= quality <= 10                                   # src/enc.rs:38
        self.ctx, self.writer, self.zero_copy = ctx, writer, zero_copy
        self.width, self.height = int(width), int(height)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.pfv_gop_encoder_create(ctx.handle, self.width, self.height, int(framerate), int(quality), int(max_gops), int(max_gop_frames),
                                                  int(payload_budget), ctypes.byref(h)))
        self.handle = h
        self.finished = False
        ctx._sessions.add(self)
        self._flush()                                               # header (src/enc.rs:70)

    def _flush(self):
        # packet headers and payloads where they lie (pfv_gop_encoder_drain_iov): one write per segment, like the reference's W: Write
        iov, n = ctypes.c_void_p(), ctypes.c_size_t()
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_drain_iov(self.handle, ctypes.byref(iov), ctypes.byref(n)))
        if not n.value:
            return
        segs = ctypes.cast(iov, ctypes.POINTER(_IoVec))
        for i in range(n.value):
            p, ln = segs[i].data, segs[i].len
            if self.zero_copy:      # a view into the library's page-locked memory, valid during this write() only
                self.writer.write(memoryview((ctypes.c_ubyte * ln).from_address(p)))
            else:
                self.writer.write(ctypes.string_at(p, ln))

    def _planes(self, frame):
        if isinstance(frame, VideoFrame):
            self._check_frame(frame)
            return ptr(frame.plane_y.pixels), ptr(frame.plane_u.pixels), ptr(frame.plane_v.pixels)
        y, u, v = frame
        assert not self.finished and y.size == self.width * self.height and u.size == v.size == (self.width // 2) * (self.height // 2)
        return ptr(y), ptr(u), ptr(v)

    def encode_iframe(self, frame):
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_encode_iframe(self.handle, *self._planes(frame)))
        self._flush()

    def encode_pframe(self, frame):
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_encode_pframe(self.handle, *self._planes(frame)))
        self._flush()

    def set_frames_by_reference(self, on: bool = True):
        """device frames (encode_*_dev) are read where they lie instead of being copied into the batch: the caller keeps each frame valid and
        unchanged until its packet has reached the writer (or flush() / finish() returned).  pfv_gop_encoder_set_frames_by_reference."""
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_set_frames_by_reference(self.handle, 1 if on else 0))

    def encode_iframe_dev(self, frame_dev: int):
        """a packed frame (Y | U | V) in DEVICE memory: pfv_gop_encoder_encode_iframe_dev.  STREAM-ORDERED on the context this encoder was created
        on, like every *_dev call: the frame is copied into the batch asynchronously on that context's stream and the call returns WITHOUT a host
        wait.  Whatever produces the frame, and whatever overwrites it afterwards, must be enqueued on the same context's stream (or ordered
        against it: Context.wait_event) -- a hipMemcpy from the host or a kernel on another stream right after the call races with the copy.
        The encoder must be closed before its context is."""
        assert not self.finished
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_encode_iframe_dev(self.handle, ctypes.c_void_p(int(frame_dev))))
        self._flush()

    def encode_pframe_dev(self, frame_dev: int):
        """see encode_iframe_dev: stream-ordered on the encoder's context, no host wait"""
        assert not self.finished
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_encode_pframe_dev(self.handle, ctypes.c_void_p(int(frame_dev))))
        self._flush()

    def encode_dropframe(self):
        assert not self.finished
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_encode_dropframe(self.handle))

    def flush(self):
        """every frame handed over so far becomes packets at the writer now"""
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_flush(self.handle))
        self._flush()

    def finish(self):
        assert not self.finished                                    # src/enc.rs:183
        self.ctx.check(self.ctx._lib.pfv_gop_encoder_finish(self.handle))
        self.finished = True
        self._flush()

    @property
    def batches(self) -> int:
        return int(self.ctx._lib.pfv_gop_encoder_batches(self.handle))

    def stats(self) -> dict:
        """host seconds so far, by what the object was waiting for (pfv_gop_encoder_stats)"""
        a = (ctypes.c_double * 7)()
        n = self.ctx._lib.pfv_gop_encoder_stats(self.handle, a, 7)
        return dict(zip(("upload_wait_s", "enqueue_s", "kernel_wait_s", "payload_download_s", "packet_assembly_s", "frames_by_reference", "batches_redone"),
                        list(a)[:n]))

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            if not self.finished:                                   # impl Drop (src/enc.rs:28-34)
                try:
                    self.finish()
                except _lib.PfvError:
                    pass                                            # a failed stream stays incomplete
            self.ctx._lib.pfv_gop_encoder_destroy(self.handle)
        self.handle = None
