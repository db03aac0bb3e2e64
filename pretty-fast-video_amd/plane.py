"""VideoPlane and the plane-level operators -- mirror of src/plane.rs and of
``impl VideoPlane`` in src/common.rs:351-521.  Names, argument meaning and error behaviour
follow the reference; the thread-pool argument is a :class:`Context`.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .context import Context, ptr


def pad16(x: int) -> int:
    return x + (16 - (x % 16)) % 16   # src/common.rs:352-353


@dataclass
class EncodedIPlane:
    """src/common.rs:37-43, flattened: blocks[i] = 4 subblocks x 64 zigzag-ordered i16."""
    width: int
    height: int
    blocks_wide: int
    blocks_high: int
    blocks: np.ndarray          # int16 [blocks_wide*blocks_high, 256]


@dataclass
class EncodedPPlane:
    """src/common.rs:45-51, flattened DeltaEncodedMacroBlock (:14-19)."""
    width: int
    height: int
    blocks_wide: int
    blocks_high: int
    motion: np.ndarray          # int8 [n, 2]  (motion_x, motion_y)
    has_coeff: np.ndarray       # uint8 [n]    (subblocks.is_some())
    blocks: np.ndarray          # int16 [n, 256]; zeros where has_coeff == 0


def _q(q_table) -> np.ndarray:
    q = np.ascontiguousarray(q_table, dtype=np.int32)
    assert q.shape == (64,)
    return q


class VideoPlane:
    """src/plane.rs:1-36."""

    def __init__(self, width: int, height: int):
        self.width = int(width)
        self.height = int(height)
        self.pixels = np.zeros(self.width * self.height, dtype=np.uint8)

    @staticmethod
    def from_slice(width: int, height: int, buffer) -> "VideoPlane":
        buf = np.asarray(buffer, dtype=np.uint8).reshape(-1)
        assert buf.size == width * height       # src/plane.rs:13
        p = VideoPlane(width, height)
        p.pixels[:] = buf
        return p

    def image(self) -> np.ndarray:
        return self.pixels.reshape(self.height, self.width)

    def blit(self, src: "VideoPlane", dx: int, dy: int, sx: int, sy: int, sw: int, sh: int):
        """src/plane.rs:20-29 (host-side container op; the device twin is pfv_blit_dev)."""
        self.image()[dy:dy + sh, dx:dx + sw] = src.image()[sy:sy + sh, sx:sx + sw]

    def get_slice(self, sx: int, sy: int, sw: int, sh: int) -> "VideoPlane":
        out = VideoPlane(sw, sh)
        out.blit(self, 0, 0, sx, sy, sw, sh)
        return out

    # ---------------------------------------------------------------- operators (run on the GPU)
    def encode_plane(self, q_table, clear_color: int, ctx: Context) -> EncodedIPlane:
        """src/common.rs:351-386."""
        pw, ph = pad16(self.width), pad16(self.height)
        bw, bh = pw // 16, ph // 16
        coef = np.empty((bw * bh, 256), dtype=np.int16)
        q = _q(q_table)
        ctx.check(ctx._lib.pfv_encode_plane(ctx.handle, ptr(self.pixels), self.width, self.height, ptr(q),
                                            int(clear_color), ptr(coef)))
        return EncodedIPlane(pw, ph, bw, bh, coef)

    def encode_plane_delta(self, refplane: "VideoPlane", q_table, px_err: float, clear_color: int,
                           ctx: Context) -> EncodedPPlane:
        """src/common.rs:388-421.  refplane must have the padded dimensions."""
        pw, ph = pad16(self.width), pad16(self.height)
        assert refplane.width == pw and refplane.height == ph
        bw, bh = pw // 16, ph // 16
        n = bw * bh
        mv = np.empty((n, 2), dtype=np.int8)
        has = np.empty(n, dtype=np.uint8)
        coef = np.empty((n, 256), dtype=np.int16)
        q = _q(q_table)
        ctx.check(ctx._lib.pfv_encode_plane_delta(ctx.handle, ptr(self.pixels), self.width, self.height,
                                                  ptr(refplane.pixels), ptr(q), float(px_err), int(clear_color),
                                                  ptr(mv), ptr(has), ptr(coef)))
        return EncodedPPlane(pw, ph, bw, bh, mv, has, coef)

    @staticmethod
    def decode_plane(src: EncodedIPlane, q_table, ctx: Context) -> "VideoPlane":
        """src/common.rs:423-446."""
        plane = VideoPlane(src.blocks_wide * 16, src.blocks_high * 16)
        VideoPlane.decode_plane_into(src, q_table, plane, ctx)
        return plane

    @staticmethod
    def decode_plane_into(src: EncodedIPlane, q_table, target: "VideoPlane", ctx: Context):
        """src/common.rs:477-496."""
        assert target.width == src.blocks_wide * 16 and target.height == src.blocks_high * 16
        q = _q(q_table)
        blocks = np.ascontiguousarray(src.blocks, dtype=np.int16)
        ctx.check(ctx._lib.pfv_decode_plane_into(ctx.handle, ptr(blocks), src.blocks_wide, src.blocks_high, ptr(q),
                                                 ptr(target.pixels)))

    @staticmethod
    def decode_plane_delta(src: EncodedPPlane, refplane: "VideoPlane", q_table, ctx: Context) -> "VideoPlane":
        """src/common.rs:448-475."""
        assert refplane.width == src.blocks_wide * 16 and refplane.height == src.blocks_high * 16
        plane = VideoPlane(src.blocks_wide * 16, src.blocks_high * 16)
        q = _q(q_table)
        mv = np.ascontiguousarray(src.motion, dtype=np.int8)
        has = np.ascontiguousarray(src.has_coeff, dtype=np.uint8)
        blocks = np.ascontiguousarray(src.blocks, dtype=np.int16)
        ctx.check(ctx._lib.pfv_decode_plane_delta(ctx.handle, ptr(mv), ptr(has), ptr(blocks), src.blocks_wide,
                                                  src.blocks_high, ptr(q), ptr(refplane.pixels), ptr(plane.pixels)))
        return plane

    @staticmethod
    def decode_plane_delta_into(src: EncodedPPlane, refplane: "VideoPlane", q_table, ctx: Context):
        """src/common.rs:498-521 (refplane is read, then overwritten)."""
        assert refplane.width == src.blocks_wide * 16 and refplane.height == src.blocks_high * 16
        q = _q(q_table)
        mv = np.ascontiguousarray(src.motion, dtype=np.int8)
        has = np.ascontiguousarray(src.has_coeff, dtype=np.uint8)
        blocks = np.ascontiguousarray(src.blocks, dtype=np.int16)
        ctx.check(ctx._lib.pfv_decode_plane_delta_into(ctx.handle, ptr(mv), ptr(has), ptr(blocks), src.blocks_wide,
                                                       src.blocks_high, ptr(q), ptr(refplane.pixels)))

    # ---------------------------------------------------------------- src/common.rs:523-556 (colour utilities, host)
    def reduce(self) -> "VideoPlane":
        out = VideoPlane(self.width // 2, self.height // 2)
        out.pixels[:] = self.image()[0:out.height * 2:2, 0:out.width * 2:2].reshape(-1)
        return out

    def double(self) -> "VideoPlane":
        out = VideoPlane(self.width * 2, self.height * 2)
        out.pixels[:] = np.repeat(np.repeat(self.image(), 2, axis=0), 2, axis=1).reshape(-1)
        return out
