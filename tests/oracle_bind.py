"""ctypes binding of the CPU oracle (oracle/libpfv_oracle.so).  TEST INFRASTRUCTURE: imported
only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_uint8, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libpfv_oracle.so")


def _p(a: np.ndarray):
    return a.ctypes.data_as(c_void_p)


def pad16(x: int) -> int:
    return x + (16 - (x % 16)) % 16


class Oracle:
    def __init__(self):
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
        if not os.path.exists(ORACLE_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
            subprocess.run(["make", "-C", ORACLE_DIR, "-B", "all"], check=True, stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(ORACLE_LIB)
        L.pfvo_encode_plane.argtypes = [c_void_p, c_int, c_int, c_void_p, c_uint8, c_void_p, c_int]
        L.pfvo_encode_plane_delta.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_uint8,
                                              c_void_p, c_void_p, c_void_p, c_int]
        L.pfvo_decode_plane_into.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]
        L.pfvo_decode_plane_delta.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int]
        L.pfvo_decode_plane_delta.restype = c_int
        L.pfvo_qtables.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_float)]
        L.pfvo_encoder_new.argtypes = [c_int, c_int, c_int, c_int]
        L.pfvo_encoder_new.restype = c_void_p
        L.pfvo_encoder_free.argtypes = [c_void_p]
        L.pfvo_encoder_total_blocks.argtypes = [c_void_p]
        L.pfvo_encoder_prev_plane.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
        L.pfvo_encoder_prev_plane.restype = c_void_p
        L.pfvo_encode_iframe.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.pfvo_encode_pframe.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.pfvo_blit.argtypes = [c_void_p, c_int, c_void_p, c_int] + [c_int] * 6
        L.pfvo_blit.restype = None
        L.pfvo_reduce.argtypes = L.pfvo_double.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.pfvo_reduce.restype = L.pfvo_double.restype = None
        self.L = L

    # ---- plane container ops (src/plane.rs:20-29, src/common.rs:523-556)
    def blit(self, dst, src, dx, dy, sx, sy, sw, sh):
        """VideoPlane::blit on 2-D u8 arrays; returns the new destination (dst is not modified)"""
        d = np.ascontiguousarray(dst, dtype=np.uint8).copy()
        s = np.ascontiguousarray(src, dtype=np.uint8)
        assert 0 <= dx and dx + sw <= d.shape[1] and 0 <= dy and dy + sh <= d.shape[0]
        assert 0 <= sx and sx + sw <= s.shape[1] and 0 <= sy and sy + sh <= s.shape[0]
        self.L.pfvo_blit(_p(d), d.shape[1], _p(s), s.shape[1], dx, dy, sx, sy, sw, sh)
        return d

    def reduce(self, src):
        s = np.ascontiguousarray(src, dtype=np.uint8)
        h, w = s.shape
        out = np.zeros((h // 2, w // 2), dtype=np.uint8)
        self.L.pfvo_reduce(_p(s), w, h, _p(out))
        return out

    def double(self, src):
        s = np.ascontiguousarray(src, dtype=np.uint8)
        h, w = s.shape
        out = np.zeros((h * 2, w * 2), dtype=np.uint8)
        self.L.pfvo_double(_p(s), w, h, _p(out))
        return out

    # ---- 1-D / subblock
    def fdct8(self, v):
        a = np.ascontiguousarray(v, dtype=np.int32).copy()
        self.L.pfvo_fdct8(_p(a))
        return a

    def idct8(self, v):
        a = np.ascontiguousarray(v, dtype=np.int32).copy()
        self.L.pfvo_idct8(_p(a))
        return a

    def encode_subblock(self, px, q):
        px = np.ascontiguousarray(px, dtype=np.uint8).reshape(64)
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros(64, dtype=np.int16)
        self.L.pfvo_encode_subblock(_p(px), _p(q), _p(out))
        return out

    def encode_subblock_delta(self, d, q):
        d = np.ascontiguousarray(d, dtype=np.int16).reshape(64)
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros(64, dtype=np.int16)
        self.L.pfvo_encode_subblock_delta(_p(d), _p(q), _p(out))
        return out

    def decode_subblock(self, c, q):
        c = np.ascontiguousarray(c, dtype=np.int16).reshape(64)
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros(64, dtype=np.uint8)
        self.L.pfvo_decode_subblock(_p(c), _p(q), _p(out))
        return out

    def qtables(self, quality: int):
        t = [np.zeros(64, dtype=np.int32) for _ in range(4)]
        pe = c_float()
        self.L.pfvo_qtables(quality, _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), ctypes.byref(pe))
        return t[0], t[1], t[2], t[3], float(pe.value)

    # ---- plane level
    def encode_plane(self, px, q, clear, threads=1):
        px = np.ascontiguousarray(px, dtype=np.uint8)
        h, w = px.shape
        q = np.ascontiguousarray(q, dtype=np.int32)
        bw, bh = pad16(w) // 16, pad16(h) // 16
        coef = np.zeros((bw * bh, 256), dtype=np.int16)
        self.L.pfvo_encode_plane(_p(px), w, h, _p(q), clear, _p(coef), threads)
        return coef, bw, bh

    def encode_plane_delta(self, px, ref, q, px_err, clear, threads=1):
        px = np.ascontiguousarray(px, dtype=np.uint8)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        h, w = px.shape
        rh, rw = ref.shape
        q = np.ascontiguousarray(q, dtype=np.int32)
        bw, bh = pad16(w) // 16, pad16(h) // 16
        n = bw * bh
        mv = np.zeros((n, 2), dtype=np.int8)
        has = np.zeros(n, dtype=np.uint8)
        coef = np.zeros((n, 256), dtype=np.int16)
        self.L.pfvo_encode_plane_delta(_p(px), w, h, _p(ref), rw, rh, _p(q), px_err, clear, _p(mv), _p(has), _p(coef), threads)
        return mv, has, coef

    def decode_plane(self, coef, bw, bh, q, threads=1):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros((bh * 16, bw * 16), dtype=np.uint8)
        self.L.pfvo_decode_plane_into(_p(coef), bw, bh, _p(q), _p(out), threads)
        return out

    def decode_plane_delta(self, mv, has, coef, bw, bh, q, ref, threads=1):
        mv = np.ascontiguousarray(mv, dtype=np.int8)
        has = np.ascontiguousarray(has, dtype=np.uint8)
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        q = np.ascontiguousarray(q, dtype=np.int32)
        out = np.zeros((bh * 16, bw * 16), dtype=np.uint8)
        rc = self.L.pfvo_decode_plane_delta(_p(mv), _p(has), _p(coef), bw, bh, _p(q), _p(ref), _p(out), threads)
        if rc != 0:
            raise ValueError("motion vector outside the reference plane")
        return out

    def encoder(self, width, height, quality, threads=1):
        return OracleEncoder(self, width, height, quality, threads)


class OracleEncoder:
    """hot-path half of enc::Encoder on the CPU oracle (src/enc.rs:84-97, 134-147)"""

    def __init__(self, ora: Oracle, width, height, quality, threads=1):
        self.L = ora.L
        self.h = self.L.pfvo_encoder_new(width, height, quality, threads)
        assert self.h
        self.width, self.height = width, height
        self.total_blocks = self.L.pfvo_encoder_total_blocks(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pfvo_encoder_free(self.h)
            self.h = None

    def _split(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        w, h = self.width, self.height
        o1, o2 = w * h, w * h + (w // 2) * (h // 2)
        return np.ascontiguousarray(f[:o1]), np.ascontiguousarray(f[o1:o2]), np.ascontiguousarray(f[o2:])

    def encode_iframe(self, frame):
        y, u, v = self._split(frame)
        coef = np.zeros((self.total_blocks, 256), dtype=np.int16)
        self.L.pfvo_encode_iframe(self.h, _p(y), _p(u), _p(v), _p(coef))
        return coef

    def encode_pframe(self, frame):
        y, u, v = self._split(frame)
        mv = np.zeros((self.total_blocks, 2), dtype=np.int8)
        has = np.zeros(self.total_blocks, dtype=np.uint8)
        coef = np.zeros((self.total_blocks, 256), dtype=np.int16)
        self.L.pfvo_encode_pframe(self.h, _p(y), _p(u), _p(v), _p(mv), _p(has), _p(coef))
        return mv, has, coef

    def prev_frame(self):
        parts = []
        for p in range(3):
            pw, ph = c_int(), c_int()
            ptr = self.L.pfvo_encoder_prev_plane(self.h, p, ctypes.byref(pw), ctypes.byref(ph))
            n = pw.value * ph.value
            parts.append(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(c_uint8)), shape=(n,)).copy())
        return np.concatenate(parts)


class OracleDecoder:
    """hot-path half of dec::Decoder on the CPU oracle (src/dec.rs:298-323, 419-445)"""

    def __init__(self, ora: Oracle, width, height, qtables, threads=1):
        self.L = ora.L
        L = self.L
        L.pfvo_decoder_new.argtypes = [c_int, c_int, c_void_p, c_int, c_int]
        L.pfvo_decoder_new.restype = c_void_p
        L.pfvo_decoder_free.argtypes = [c_void_p]
        L.pfvo_decoder_plane.argtypes = [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
        L.pfvo_decoder_plane.restype = c_void_p
        L.pfvo_decode_iframe.argtypes = [c_void_p, c_void_p, c_void_p]
        L.pfvo_decode_pframe.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.pfvo_decode_pframe.restype = c_int
        q = np.ascontiguousarray(qtables, dtype=np.int32).reshape(-1, 64)
        self.h = L.pfvo_decoder_new(width, height, _p(q), q.shape[0], threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pfvo_decoder_free(self.h)
            self.h = None

    def decode_iframe(self, coef, qidx=(0, 1, 1)):
        c = np.ascontiguousarray(coef, dtype=np.int16)
        qi = np.ascontiguousarray(qidx, dtype=np.uint8)
        self.L.pfvo_decode_iframe(self.h, _p(c), _p(qi))

    def decode_pframe(self, mv, has, coef, qidx=(2, 3, 3)):
        m = np.ascontiguousarray(mv, dtype=np.int8)
        hc = np.ascontiguousarray(has, dtype=np.uint8)
        c = np.ascontiguousarray(coef, dtype=np.int16)
        qi = np.ascontiguousarray(qidx, dtype=np.uint8)
        if self.L.pfvo_decode_pframe(self.h, _p(m), _p(hc), _p(c), _p(qi)) != 0:
            raise ValueError("motion vector outside the reference plane")

    def framebuffer(self):
        parts = []
        for p in range(3):
            pw, ph = c_int(), c_int()
            ptr = self.L.pfvo_decoder_plane(self.h, p, ctypes.byref(pw), ctypes.byref(ph))
            parts.append(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(c_uint8)), shape=(pw.value * ph.value,)).copy())
        return np.concatenate(parts)


class OracleStreamEncoder:
    """enc::Encoder on the CPU oracle, whole .pfv byte stream (oracle/pfv_oracle_entropy.c)"""

    def __init__(self, ora: Oracle, width, height, framerate, quality, threads=1):
        L = self.L = ora.L
        L.pfvo_stream_encoder_new.argtypes = [c_int] * 5
        L.pfvo_stream_encoder_new.restype = c_void_p
        for name in ("pfvo_stream_encode_iframe", "pfvo_stream_encode_pframe"):
            getattr(L, name).argtypes = [c_void_p] * 4
        L.pfvo_stream_encode_dropframe.argtypes = [c_void_p]
        L.pfvo_stream_finish.argtypes = [c_void_p]
        L.pfvo_stream_bytes.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        L.pfvo_stream_bytes.restype = c_void_p
        L.pfvo_stream_encoder_free.argtypes = [c_void_p]
        self.h = L.pfvo_stream_encoder_new(width, height, framerate, quality, threads)
        assert self.h
        self.width, self.height = width, height

    def _split(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        w, h = self.width, self.height
        o1, o2 = w * h, w * h + (w // 2) * (h // 2)
        return np.ascontiguousarray(f[:o1]), np.ascontiguousarray(f[o1:o2]), np.ascontiguousarray(f[o2:])

    def encode_iframe(self, frame):
        y, u, v = self._split(frame)
        self.L.pfvo_stream_encode_iframe(self.h, _p(y), _p(u), _p(v))

    def encode_pframe(self, frame):
        y, u, v = self._split(frame)
        self.L.pfvo_stream_encode_pframe(self.h, _p(y), _p(u), _p(v))

    def encode_dropframe(self):
        self.L.pfvo_stream_encode_dropframe(self.h)

    def finish(self):
        self.L.pfvo_stream_finish(self.h)

    def bytes(self) -> bytes:
        n = ctypes.c_size_t()
        p = self.L.pfvo_stream_bytes(self.h, ctypes.byref(n))
        return ctypes.string_at(p, n.value)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pfvo_stream_encoder_free(self.h)
            self.h = None


class OracleStreamDecoder:
    """dec::Decoder on the CPU oracle"""

    def __init__(self, ora: Oracle, data: bytes, threads=1):
        L = self.L = ora.L
        L.pfvo_stream_decoder_new.argtypes = [c_void_p, ctypes.c_size_t, c_int, ctypes.POINTER(c_int)]
        L.pfvo_stream_decoder_new.restype = c_void_p
        L.pfvo_stream_decoder_free.argtypes = [c_void_p]
        L.pfvo_stream_decoder_info.argtypes = [c_void_p] + [ctypes.POINTER(c_int)] * 3
        L.pfvo_stream_decoder_reset.argtypes = [c_void_p]
        L.pfvo_stream_advance_frame.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_int)]
        L.pfvo_stream_advance_frame.restype = c_int
        self._data = np.frombuffer(data, dtype=np.uint8).copy()
        err = c_int()
        self.h = L.pfvo_stream_decoder_new(_p(self._data), self._data.size, threads, ctypes.byref(err))
        self.err = err.value
        if self.h:
            w, h, f = c_int(), c_int(), c_int()
            L.pfvo_stream_decoder_info(self.h, ctypes.byref(w), ctypes.byref(h), ctypes.byref(f))
            self.width, self.height, self.framerate = w.value, h.value, f.value
            self._frame = np.zeros(self.width * self.height + 2 * (self.width // 2) * (self.height // 2), dtype=np.uint8)

    def advance_frame(self):
        """-> (rc, frame or None): rc 1 more data / 0 EOF / negative error"""
        got = c_int()
        rc = self.L.pfvo_stream_advance_frame(self.h, _p(self._frame), ctypes.byref(got))
        return rc, (self._frame.copy() if got.value else None)

    def reset(self):
        self.L.pfvo_stream_decoder_reset(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pfvo_stream_decoder_free(self.h)
            self.h = None
