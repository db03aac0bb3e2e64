"""Device-resident encoder / decoder sessions: the hot-path half of ``enc::Encoder`` and
``dec::Decoder`` (src/enc.rs:12-173, src/dec.rs:15-224) for ``n_streams`` independent
streams per launch.  prev_frame / framebuffer stay in HBM between frames.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .context import Context, ptr


def qtables_from_quality(quality: int):
    """Encoder::new's table derivation (src/enc.rs:40-51) -> (intra_l, intra_c, inter_l, inter_c, px_err)."""
    lib = _lib.load()
    t = [np.zeros(64, dtype=np.int32) for _ in range(4)]
    pe = ctypes.c_float()
    rc = lib.pfv_qtables_from_quality(int(quality), ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ctypes.byref(pe))
    _lib.check(None, rc)
    return t[0], t[1], t[2], t[3], float(pe.value)


class _Geometry:
    def __init__(self, width: int, height: int, n_streams: int):
        lib = _lib.load()
        self.width, self.height, self.n_streams = int(width), int(height), int(n_streams)
        self.frame_bytes = int(lib.pfv_frame_bytes(width, height))
        self.padded_frame_bytes = int(lib.pfv_padded_frame_bytes(width, height))
        self.total_blocks = int(lib.pfv_total_blocks(width, height))


class EncoderSession(_Geometry):
    def __init__(self, ctx: Context, width: int, height: int, quality: int, n_streams: int = 1):
        super().__init__(width, height, n_streams)
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.pfv_enc_session_create(ctx.handle, int(width), int(height), int(quality), int(n_streams),
                                                  ctypes.byref(h)))
        self.handle = h
        ctx._sessions.add(self)

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.pfv_enc_session_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # host-buffer forms ------------------------------------------------
    def _frames(self, frames) -> np.ndarray:
        f = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1)
        assert f.size == self.frame_bytes * self.n_streams
        return f

    def encode_iframe(self, frames) -> np.ndarray:
        """src/enc.rs:84-97 for every stream; returns coef int16 [n_streams, total_blocks, 256]."""
        f = self._frames(frames)
        coef = np.empty((self.n_streams, self.total_blocks, 256), dtype=np.int16)
        self.ctx.check(self.ctx._lib.pfv_enc_iframe(self.handle, ptr(f), ptr(coef)))
        return coef

    def encode_pframe(self, frames):
        """src/enc.rs:134-147 for every stream; returns (mv, has_coef, coef)."""
        f = self._frames(frames)
        mv = np.empty((self.n_streams, self.total_blocks, 2), dtype=np.int8)
        has = np.empty((self.n_streams, self.total_blocks), dtype=np.uint8)
        coef = np.empty((self.n_streams, self.total_blocks, 256), dtype=np.int16)
        self.ctx.check(self.ctx._lib.pfv_enc_pframe(self.handle, ptr(f), ptr(mv), ptr(has), ptr(coef)))
        return mv, has, coef

    def prev_frame(self) -> np.ndarray:
        out = np.empty((self.n_streams, self.padded_frame_bytes), dtype=np.uint8)
        self.ctx.check(self.ctx._lib.pfv_enc_prev_frame(self.handle, ptr(out)))
        return out

    # device-pointer forms (asynchronous on the context's stream) -------
    def encode_iframe_dev(self, frames_dev: int, coef_dev: int):
        self.ctx.check(self.ctx._lib.pfv_enc_iframe_dev(self.handle, ctypes.c_void_p(frames_dev), ctypes.c_void_p(coef_dev)))

    def encode_pframe_dev(self, frames_dev: int, mv_dev: int, has_dev: int, coef_dev: int):
        self.ctx.check(self.ctx._lib.pfv_enc_pframe_dev(self.handle, ctypes.c_void_p(frames_dev), ctypes.c_void_p(mv_dev),
                                                        ctypes.c_void_p(has_dev), ctypes.c_void_p(coef_dev)))

    # GOP-batched use: the slots hold the GOPs of one stream (include/pfv_hip.h, pfv_enc_session_set_window) -----------
    def set_frame_stride(self, stride_bytes: int = 0):
        self.ctx.check(self.ctx._lib.pfv_enc_session_set_frame_stride(self.handle, int(stride_bytes)))

    def set_window(self, first: int = 0, count: int | None = None):
        self.ctx.check(self.ctx._lib.pfv_enc_session_set_window(self.handle, int(first), int(self.n_streams - first if count is None else count)))

    # device entropy stage (RLE + Huffman + bit packing of enc.rs:237-470 on the device) ---------------
    def enable_entropy(self, payload_cap: int = 0, async_stream: bool = False):
        """allocate the stage; payload_cap = bytes per stream (0: worst case for the geometry).  async_stream: run it
        on its own HIP stream (the caller then alternates between two sets of encode output buffers)"""
        self.ctx.check(self.ctx._lib.pfv_enc_entropy_enable(self.handle, int(payload_cap)))
        self.ctx.check(self.ctx._lib.pfv_enc_entropy_set_async(self.handle, 1 if async_stream else 0))

    def entropy_join(self):
        """the context's stream waits for the entropy stage (no host synchronisation)"""
        self.ctx.check(self.ctx._lib.pfv_enc_entropy_join(self.handle))

    def pack_iframe_dev(self, coef_dev: int):
        self.ctx.check(self.ctx._lib.pfv_enc_pack_iframe_dev(self.handle, ctypes.c_void_p(coef_dev)))

    def pack_pframe_dev(self, mv_dev: int, has_dev: int, coef_dev: int):
        self.ctx.check(self.ctx._lib.pfv_enc_pack_pframe_dev(self.handle, ctypes.c_void_p(mv_dev), ctypes.c_void_p(has_dev),
                                                             ctypes.c_void_p(coef_dev)))

    def payload_sizes(self) -> np.ndarray:
        """bytes per stream of the last pack call (synchronises); raises PfvError on oversize coefficients / capacity"""
        sizes = np.zeros(self.n_streams, dtype=np.uint32)
        self.ctx.check(self.ctx._lib.pfv_enc_payload_sizes(self.handle, ptr(sizes)))
        return sizes

    def payload(self, stream: int, nbytes: int) -> bytes:
        out = np.empty(max(int(nbytes), 1), dtype=np.uint8)
        self.ctx.check(self.ctx._lib.pfv_enc_payload_fetch(self.handle, int(stream), ptr(out), int(nbytes)))
        return out[:nbytes].tobytes()

    def payloads(self, out: np.ndarray):
        """all streams' payloads of the last pack call with one device-to-host copy into `out` (uint8, ideally from
        Context.host_array); returns (sizes, offsets): stream s is out[offsets[s] : offsets[s] + sizes[s]]"""
        sizes = np.zeros(self.n_streams, dtype=np.uint32)
        offsets = np.zeros(self.n_streams, dtype=np.uint64)
        self.ctx.check(self.ctx._lib.pfv_enc_payloads_fetch(self.handle, ptr(out), out.size, ptr(sizes), ptr(offsets)))
        return sizes, offsets

    def payload_dev(self, stream: int) -> int:
        return int(self.ctx._lib.pfv_enc_payload_dev(self.handle, int(stream)) or 0)


class DecoderSession(_Geometry):
    def __init__(self, ctx: Context, width: int, height: int, qtables, n_streams: int = 1):
        super().__init__(width, height, n_streams)
        self.ctx = ctx
        q = np.ascontiguousarray(qtables, dtype=np.int32).reshape(-1, 64)
        h = ctypes.c_void_p()
        ctx.check(ctx._lib.pfv_dec_session_create(ctx.handle, int(width), int(height), ptr(q), int(q.shape[0]),
                                                  int(n_streams), ctypes.byref(h)))
        self.handle = h
        ctx._sessions.add(self)

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.pfv_dec_session_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _qidx(qidx) -> np.ndarray:
        q = np.ascontiguousarray(qidx, dtype=np.uint8)
        assert q.shape == (3,)
        return q

    def decode_iframe(self, coef, qidx=(0, 1, 1)):
        c = np.ascontiguousarray(coef, dtype=np.int16)
        assert c.size == self.n_streams * self.total_blocks * 256
        self.ctx.check(self.ctx._lib.pfv_dec_iframe(self.handle, ptr(c), ptr(self._qidx(qidx))))

    def decode_iframe_sparse(self, idx, val, qidx=(0, 1, 1)):
        """non-zero coefficients as (flat index into [stream][macroblock][256], value) pairs; the rest is zero"""
        i = np.ascontiguousarray(idx, dtype=np.uint32)
        v = np.ascontiguousarray(val, dtype=np.int16)
        assert i.shape == v.shape and i.ndim == 1
        self.ctx.check(self.ctx._lib.pfv_dec_iframe_sparse(self.handle, ptr(i), ptr(v), i.size, ptr(self._qidx(qidx))))

    def decode_pframe_sparse(self, mv, has_coef, idx, val, qidx=(2, 3, 3)):
        m = np.ascontiguousarray(mv, dtype=np.int8)
        hc = np.ascontiguousarray(has_coef, dtype=np.uint8)
        i = np.ascontiguousarray(idx, dtype=np.uint32)
        v = np.ascontiguousarray(val, dtype=np.int16)
        assert i.shape == v.shape and i.ndim == 1
        self.ctx.check(self.ctx._lib.pfv_dec_pframe_sparse(self.handle, ptr(m), ptr(hc), ptr(i), ptr(v), i.size, ptr(self._qidx(qidx))))

    def decode_pframe(self, mv, has_coef, coef, qidx=(2, 3, 3)):
        m = np.ascontiguousarray(mv, dtype=np.int8)
        hc = np.ascontiguousarray(has_coef, dtype=np.uint8)
        c = np.ascontiguousarray(coef, dtype=np.int16)
        assert c.size == self.n_streams * self.total_blocks * 256
        self.ctx.check(self.ctx._lib.pfv_dec_pframe(self.handle, ptr(m), ptr(hc), ptr(c), ptr(self._qidx(qidx))))

    def coef_lists(self, coef, has_coef=None):
        """dense coefficients [n_streams][total_blocks][256] -> (entries per stream, counts [n_streams][total_blocks + 1]): the
        coefficient-list form of pfv_dec_*_lists_dev (pfv_coef_lists_from_dense)"""
        c = np.ascontiguousarray(coef, dtype=np.int16).reshape(self.n_streams, self.total_blocks * 256)
        hc = None if has_coef is None else np.ascontiguousarray(has_coef, dtype=np.uint8).reshape(self.n_streams, self.total_blocks)
        entries, counts = [], np.zeros((self.n_streams, self.total_blocks + 1), dtype=np.uint32)
        for k in range(self.n_streams):
            e = np.empty(self.total_blocks * 256, dtype=np.uint32)
            n = ctypes.c_size_t(0)
            rc = self.ctx._lib.pfv_coef_lists_from_dense(ptr(c[k]), ptr(hc[k]) if hc is not None else None, self.total_blocks, ptr(e), e.size,
                                                         ptr(counts[k]), ctypes.byref(n))
            assert rc == 0, rc
            entries.append(e[:n.value].copy())
        return entries, counts

    def _upload_lists(self, entries, counts):
        """entries of all slots back to back in one device buffer + the table of list pointers + the counts; returns the three device
        addresses (the caller frees them)"""
        ctx = self.ctx
        sizes = [max(int(e.size), 1) for e in entries]
        ent_dev = ctx.alloc(4 * sum(sizes))
        ptrs = np.zeros(self.n_streams, dtype=np.uint64)
        off = 0
        for k, e in enumerate(entries):
            ptrs[k] = ent_dev + 4 * off
            if e.size:
                ctx.upload(ent_dev + 4 * off, np.ascontiguousarray(e, dtype=np.uint32))
            off += sizes[k]
        ptr_dev = ctx.alloc(8 * self.n_streams)
        ctx.upload(ptr_dev, ptrs)
        r = np.ascontiguousarray(counts, dtype=np.uint32)
        rng_dev = ctx.alloc(r.nbytes)
        ctx.upload(rng_dev, r)
        return ent_dev, ptr_dev, rng_dev

    def decode_iframe_lists(self, entries, counts, qidx=(0, 1, 1)):
        """i-frame from coefficient lists (coef_lists): same framebuffer as decode_iframe on the dense array"""
        bufs = self._upload_lists(entries, counts)
        try:
            self.ctx.check(self.ctx._lib.pfv_dec_iframe_lists_dev(self.handle, ctypes.c_void_p(bufs[1]), ctypes.c_void_p(bufs[2]), ptr(self._qidx(qidx))))
            self.ctx.sync()
        finally:
            for b in bufs:
                self.ctx.free(b)

    def decode_pframe_lists(self, mv, has_coef, entries, counts, qidx=(2, 3, 3)):
        ctx = self.ctx
        m = np.ascontiguousarray(mv, dtype=np.int8)
        hc = np.ascontiguousarray(has_coef, dtype=np.uint8)
        bufs = list(self._upload_lists(entries, counts))
        try:
            mv_dev = ctx.alloc(m.nbytes); bufs.append(mv_dev)
            has_dev = ctx.alloc(hc.nbytes); bufs.append(has_dev)
            ctx.upload(mv_dev, m)
            ctx.upload(has_dev, hc)
            ctx.check(ctx._lib.pfv_dec_pframe_lists_dev(self.handle, ctypes.c_void_p(mv_dev), ctypes.c_void_p(has_dev), ctypes.c_void_p(bufs[1]),
                                                        ctypes.c_void_p(bufs[2]), ptr(self._qidx(qidx))))
            self.check()
        finally:
            for b in bufs:
                ctx.free(b)

    def decode_iframe_dev(self, coef_dev: int, qidx=(0, 1, 1)):
        self.ctx.check(self.ctx._lib.pfv_dec_iframe_dev(self.handle, ctypes.c_void_p(coef_dev), ptr(self._qidx(qidx))))

    def decode_pframe_dev(self, mv_dev: int, has_dev: int, coef_dev: int, qidx=(2, 3, 3)):
        self.ctx.check(self.ctx._lib.pfv_dec_pframe_dev(self.handle, ctypes.c_void_p(mv_dev), ctypes.c_void_p(has_dev),
                                                        ctypes.c_void_p(coef_dev), ptr(self._qidx(qidx))))

    def check(self):
        self.ctx.check(self.ctx._lib.pfv_dec_check(self.handle))

    def get_frame(self) -> np.ndarray:
        """retframe of every stream (src/dec.rs:195-197): uint8 [n_streams, frame_bytes]."""
        out = np.empty((self.n_streams, self.frame_bytes), dtype=np.uint8)
        self.ctx.check(self.ctx._lib.pfv_dec_get_frame(self.handle, ptr(out)))
        return out

    def set_output_dev(self, frames_dev):
        """fuse the retframe crop into the decode kernels (None switches it off)"""
        self.ctx.check(self.ctx._lib.pfv_dec_set_output_dev(self.handle, ctypes.c_void_p(frames_dev or 0)))

    def set_output_strided_dev(self, frames_dev, stride_bytes: int):
        """retframes of consecutive slots `stride_bytes` apart (GOP-batched decode: the stream appears in display order)"""
        self.ctx.check(self.ctx._lib.pfv_dec_set_output_strided_dev(self.handle, ctypes.c_void_p(frames_dev or 0), int(stride_bytes)))

    def set_window(self, first: int = 0, count: int | None = None):
        self.ctx.check(self.ctx._lib.pfv_dec_session_set_window(self.handle, int(first), int(self.n_streams - first if count is None else count)))

    def get_frame_dev(self, frames_dev: int):
        self.ctx.check(self.ctx._lib.pfv_dec_get_frame_dev(self.handle, ctypes.c_void_p(frames_dev)))

    def framebuffer(self) -> np.ndarray:
        out = np.empty((self.n_streams, self.padded_frame_bytes), dtype=np.uint8)
        self.ctx.check(self.ctx._lib.pfv_dec_framebuffer(self.handle, ptr(out)))
        return out
