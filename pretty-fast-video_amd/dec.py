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
    """dec::Decoder (src/dec.rs:15-224) over ``pfv_decoder``.  ``entropy``: where the run streams of a packet are read -- ``"auto"`` (on the
    device for payloads of 64 KiB and more), ``"host"``, ``"device"`` (every packet) -- ``None`` = as the context is set (PFV_OPT_ENTROPY_DECODE)."""
    ENTROPY = {"auto": _lib.PFV_ENTROPY_DECODE_AUTO, "host": _lib.PFV_ENTROPY_DECODE_HOST, "device": _lib.PFV_ENTROPY_DECODE_DEVICE}

    def __init__(self, reader, ctx: Context, lookahead: int | None = None, entropy=None):
        data = reader.read() if hasattr(reader, "read") else bytes(reader)
        self._data = np.frombuffer(data, dtype=np.uint8).copy()     # must outlive the native decoder
        self.ctx = ctx
        h = ctypes.c_void_p()
        before = ctx.get_option(_lib.PFV_OPT_ENTROPY_DECODE)
        if entropy is not None:
            ctx.set_option(_lib.PFV_OPT_ENTROPY_DECODE, self.ENTROPY[entropy])
        try:
            rc = ctx._lib.pfv_decoder_create(ctx.handle, self._data.ctypes.data_as(ctypes.c_void_p), self._data.size, ctypes.byref(h))
        finally:
            ctx.set_option(_lib.PFV_OPT_ENTROPY_DECODE, before)
        if rc != _lib.PFV_OK:
            msg = ctx._lib.pfv_last_error(ctx.handle)
            raise DecodeError(rc, msg.decode() if msg else "")
        self.handle = h
        ctx._sessions.add(self)
        if lookahead is not None:                                   # packets parsed ahead on this many worker threads
            ctx.check(ctx._lib.pfv_decoder_set_lookahead(h, int(lookahead)))

    def set_output_device(self, on: bool = True):
        """frames stay in device memory: ``onvideo`` then gets the packed frame's DEVICE address (an int; planes back to back),
        valid until the next advance call (pfv_decoder_set_output_device)"""
        self.ctx.check(self.ctx._lib.pfv_decoder_set_output_device(self.handle, 1 if on else 0))
        self._device_out = bool(on)

    def entropy_counts(self) -> dict:
        """packets whose run streams the device read / that its stage left to the host parser (pfv_decoder_entropy_counts)"""
        a = (ctypes.c_long * 2)()
        self.ctx._lib.pfv_decoder_entropy_counts(self.handle, a)
        return {"packets_read_on_device": int(a[0]), "packets_left_to_host_parser": int(a[1])}

    def width(self) -> int:
        return self.ctx._lib.pfv_decoder_width(self.handle)

    def height(self) -> int:
        return self.ctx._lib.pfv_decoder_height(self.handle)

    def framerate(self) -> int:
        return self.ctx._lib.pfv_decoder_framerate(self.handle)

    def reset(self):
        self.ctx.check(self.ctx._lib.pfv_decoder_reset(self.handle))

    def _callback(self, onvideo):
        if getattr(self, "_device_out", False):
            return _CB(lambda _user, y, u, v, w, h: onvideo(int(y or 0)))

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


class GopDecoder(Decoder):
    """:class:`Decoder` for one stream with the independent GOPs of the stream as the slots of every kernel launch
    (``pfv_gop_decoder``, include/pfv_hip.h): same calls, frames, order and errors; ``threads`` packet parsers work beside the caller.
    ``raw=True`` hands ``onvideo`` the three planes as zero-copy uint8 views (valid until the next batch starts) instead of a
    :class:`VideoFrame` copy.  ``entropy``: where packet payloads are read -- ``"auto"`` / ``"host"`` / ``"device"``
    (PFV_OPT_ENTROPY_DECODE, include/pfv_hip.h), ``None`` = whatever the context is set to.  ``output="device"`` (needs ``raw``): the
    frames stay in HBM and ``onvideo`` gets the three planes' device addresses (ints) instead (pfv_gop_decoder_set_output_device)."""

    def __init__(self, reader, ctx: Context, max_gops: int = 8, max_gop_frames: int = 15, threads: int = 8, raw: bool = False, entropy=None, output: str = "host", entropy_shape=None):
        data = reader.read() if hasattr(reader, "read") else bytes(reader)
        self._data = np.frombuffer(data, dtype=np.uint8).copy()     # must outlive the native decoder
        self.ctx, self.raw = ctx, raw
        h = ctypes.c_void_p()
        # entropy_shape: (lane bits, launches, inner rounds) of the device stage, None entries = as the context has them (PFV_OPT_ENTDEC_*)
        opts = [_lib.PFV_OPT_ENTROPY_DECODE, _lib.PFV_OPT_ENTDEC_LANE_BITS, _lib.PFV_OPT_ENTDEC_LAUNCHES, _lib.PFV_OPT_ENTDEC_INNER_ROUNDS]
        before = [ctx.get_option(o) for o in opts]
        wanted = [None if entropy is None else self.ENTROPY[entropy]] + list(entropy_shape or (None, None, None))
        try:
            for o, v in zip(opts, wanted):
                if v is not None:
                    ctx.set_option(o, v)
            rc = ctx._lib.pfv_gop_decoder_create(ctx.handle, self._data.ctypes.data_as(ctypes.c_void_p), self._data.size, int(max_gops), int(max_gop_frames),
                                                 int(threads), ctypes.byref(h))
        finally:
            for o, v in zip(opts, before):
                ctx.set_option(o, v)
        if rc != _lib.PFV_OK:
            msg = ctx._lib.pfv_last_error(ctx.handle)
            raise DecodeError(rc, msg.decode() if msg else "")
        self.handle = h
        ctx._sessions.add(self)
        self.output = output
        if output == "device":
            assert raw, "device output hands over addresses: raw=True"
            ctx.check(ctx._lib.pfv_gop_decoder_set_output_device(h, 1))

    def width(self) -> int:
        return self.ctx._lib.pfv_gop_decoder_width(self.handle)

    def height(self) -> int:
        return self.ctx._lib.pfv_gop_decoder_height(self.handle)

    def framerate(self) -> int:
        return self.ctx._lib.pfv_gop_decoder_framerate(self.handle)

    @property
    def batches(self) -> int:
        return int(self.ctx._lib.pfv_gop_decoder_batches(self.handle))

    def stats(self) -> dict:
        """host seconds so far, by what the object was waiting for (pfv_gop_decoder_stats)"""
        a = (ctypes.c_double * 11)()
        n = self.ctx._lib.pfv_gop_decoder_stats(self.handle, a, 11)
        out = dict(zip(("scan_s", "parse_wait_s", "device_wait_s", "enqueue_s", "final_wait_s", "device_entropy_wait_s", "packets_read_on_device",
                        "packets_left_to_host_parser", "left_unsettled", "left_irregular", "lists_spilled"), list(a)[:n]))
        for k in ("packets_read_on_device", "packets_left_to_host_parser", "left_unsettled", "left_irregular", "lists_spilled"):
            if k in out:
                out[k] = int(out[k])
        return out

    def reset(self):
        self.ctx.check(self.ctx._lib.pfv_gop_decoder_reset(self.handle))

    def _callback(self, onvideo):
        if not self.raw:
            return super()._callback(onvideo)

        if self.output == "device":
            def cbd(_user, y, u, v, w, h):
                onvideo(*(int(p or 0) for p in (y, u, v)))
            return _CB(cbd)

        def cb(_user, y, u, v, w, h):
            # the three planes of a frame lie back to back in the decoder's staging (include/pfv_hip.h): one view, three slices
            ny, nc = w * h, (w // 2) * (h // 2)
            a = np.frombuffer((ctypes.c_uint8 * (ny + 2 * nc)).from_address(y), dtype=np.uint8)
            onvideo(a[:ny], a[ny:ny + nc], a[ny + nc:])
        return _CB(cb)

    def _cached_callback(self, onvideo):
        # one ctypes trampoline per consumer, not per call (building one costs more than decoding a frame)
        if getattr(self, "_cb_for", None) is not onvideo:
            self._cb, self._cb_for = self._callback(onvideo), onvideo
        return self._cb

    def advance_frame(self, onvideo) -> bool:
        cb = self._cached_callback(onvideo)
        rc = self.ctx._lib.pfv_gop_decoder_advance_frame(self.handle, cb, None)
        if rc < 0:
            self.ctx.check(rc)
        return rc == 1

    def advance_delta(self, delta: float, onvideo) -> bool:
        cb = self._callback(onvideo)
        rc = self.ctx._lib.pfv_gop_decoder_advance_delta(self.handle, float(delta), cb, None)
        if rc < 0:
            self.ctx.check(rc)
        return rc == 1

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.pfv_gop_decoder_destroy(self.handle)
        self.handle = None


class BatchDecoder:
    """``n`` ``.pfv`` streams of one geometry and one frame-type pattern (e.g. the outputs of :class:`BatchEncoder`)
    decoded together by the C++ ``pfv_batch_decoder``: the packets of a step are bit-parsed on a worker pool (one task per
    stream), their non-zero coefficients reach the device as lists, ONE kernel launch decodes all streams, and the parse of
    step t+1 overlaps the device work of step t.  This class marshals.

    ``advance_frames()`` returns the ``[n, frame_bytes]`` array of decoded frames (a view into page-locked memory, valid
    until the call after next), ``None`` for a step of drop frames, or ``False`` at the end of the streams."""

    def __init__(self, streams, ctx: Context, threads: int = 8, entropy=None):
        """``entropy``: where the run streams of the packets are read, as for :class:`Decoder` (PFV_OPT_ENTROPY_DECODE)"""
        self.ctx = ctx
        self.data = [np.frombuffer(bytes(s.read() if hasattr(s, "read") else s), dtype=np.uint8) for s in streams]
        self.n = len(self.data)
        ptrs = (ctypes.c_void_p * self.n)(*[d.ctypes.data for d in self.data])
        lens = (ctypes.c_size_t * self.n)(*[d.size for d in self.data])
        h = ctypes.c_void_p()
        before = ctx.get_option(_lib.PFV_OPT_ENTROPY_DECODE)
        if entropy is not None:
            ctx.set_option(_lib.PFV_OPT_ENTROPY_DECODE, Decoder.ENTROPY[entropy])
        try:
            rc = ctx._lib.pfv_batch_decoder_create(ctx.handle, ptrs, lens, self.n, int(threads), ctypes.byref(h))
        finally:
            ctx.set_option(_lib.PFV_OPT_ENTROPY_DECODE, before)
        if rc:
            msg = ctx._lib.pfv_last_error(ctx.handle)
            raise DecodeError(rc, msg.decode() if msg else "")
        self.handle = h
        lib = ctx._lib
        self.width, self.height, self.framerate = (lib.pfv_batch_decoder_width(h), lib.pfv_batch_decoder_height(h),
                                                   lib.pfv_batch_decoder_framerate(h))
        self.frame_bytes = int(lib.pfv_frame_bytes(self.width, self.height))
        ctx._sessions.add(self)

    def advance_frames(self):
        out = ctypes.c_void_p()
        rc = self.ctx._lib.pfv_batch_decoder_advance(self.handle, ctypes.byref(out))
        if rc < 0:
            msg = self.ctx._lib.pfv_last_error(self.ctx.handle)
            raise DecodeError(rc, msg.decode() if msg else "")
        if rc == 0:
            return False
        if rc == 2:
            return None
        return np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint8)), shape=(self.n, self.frame_bytes))

    def entropy_counts(self) -> dict:
        """packets whose run streams the device read / that its stage left to the host parser (pfv_batch_decoder_entropy_counts)"""
        a = (ctypes.c_long * 2)()
        self.ctx._lib.pfv_batch_decoder_entropy_counts(self.handle, a)
        return {"packets_read_on_device": int(a[0]), "packets_left_to_host_parser": int(a[1])}

    @property
    def dense_steps(self) -> int:
        """steps whose coefficient lists overflowed and took the dense path"""
        return int(self.ctx._lib.pfv_batch_decoder_dense_steps(self.handle))

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.pfv_batch_decoder_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
