"""VideoFrame -- mirror of src/frame.rs:3-59 (Y plane + half-resolution U and V planes)."""
from __future__ import annotations

import numpy as np

from .plane import VideoPlane, pad16


class VideoFrame:
    def __init__(self, width: int, height: int, plane_y: VideoPlane, plane_u: VideoPlane, plane_v: VideoPlane):
        self.width, self.height = int(width), int(height)
        self.plane_y, self.plane_u, self.plane_v = plane_y, plane_u, plane_v

    @staticmethod
    def new(width: int, height: int) -> "VideoFrame":
        """src/frame.rs:12-26."""
        assert width % 2 == 0 and height % 2 == 0
        u, v = VideoPlane(width // 2, height // 2), VideoPlane(width // 2, height // 2)
        u.pixels.fill(128)
        v.pixels.fill(128)
        return VideoFrame(width, height, VideoPlane(width, height), u, v)

    @staticmethod
    def new_padded(width: int, height: int) -> "VideoFrame":
        """src/frame.rs:28-49: chroma padded from (w/2, h/2) independently."""
        cw, ch = pad16(width // 2), pad16(height // 2)
        u, v = VideoPlane(cw, ch), VideoPlane(cw, ch)
        u.pixels.fill(128)
        v.pixels.fill(128)
        return VideoFrame(width, height, VideoPlane(pad16(width), pad16(height)), u, v)

    @staticmethod
    def from_planes(width: int, height: int, plane_y: VideoPlane, plane_u: VideoPlane, plane_v: VideoPlane) -> "VideoFrame":
        """src/frame.rs:51-59: full-resolution chroma is point-sampled 2x (VideoPlane::reduce)."""
        for p in (plane_y, plane_u, plane_v):
            assert p.width == width and p.height == height
        return VideoFrame(width, height, plane_y, plane_u.reduce(), plane_v.reduce())

    # packing used by the session API: Y | U | V, tightly packed
    # ---------------------------------------------------------------- src/lib.rs:337-394 (load_frame / save_frame)
    @staticmethod
    def from_rgb(ctx, rgb: np.ndarray) -> "VideoFrame":
        """interleaved RGB8 [h, w, 3] -> 4:2:0 frame (JPEG-conversion YCbCr in f32, chroma point-sampled), on the device"""
        import ctypes
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w = rgb.shape[:2]
        assert rgb.shape == (h, w, 3) and w % 2 == 0 and h % 2 == 0
        nbytes = w * h + 2 * (w // 2) * (h // 2)
        d_rgb, d_frame = ctx.alloc(rgb.size), ctx.alloc(nbytes)
        try:
            ctx.upload(d_rgb, rgb)
            ctx.check(ctx._lib.pfv_rgb_to_yuv420_dev(ctx.handle, ctypes.c_void_p(d_rgb), w, h, ctypes.c_void_p(d_frame)))
            out = np.empty(nbytes, dtype=np.uint8)
            ctx.download(out, d_frame)
        finally:
            ctx.free(d_rgb)
            ctx.free(d_frame)
        return VideoFrame.from_packed(w, h, out)

    def to_rgb(self, ctx) -> np.ndarray:
        """4:2:0 frame -> interleaved RGB8 [h, w, 3] (chroma doubled, JPEG-conversion YCbCr in f32), on the device"""
        import ctypes
        w, h = self.width, self.height
        buf = self.packed()
        d_rgb, d_frame = ctx.alloc(w * h * 3), ctx.alloc(buf.size)
        try:
            ctx.upload(d_frame, buf)
            ctx.check(ctx._lib.pfv_yuv420_to_rgb_dev(ctx.handle, ctypes.c_void_p(d_frame), w, h, ctypes.c_void_p(d_rgb)))
            out = np.empty((h, w, 3), dtype=np.uint8)
            ctx.download(out, d_rgb)
        finally:
            ctx.free(d_rgb)
            ctx.free(d_frame)
        return out

    def packed(self) -> np.ndarray:
        return np.concatenate([self.plane_y.pixels, self.plane_u.pixels, self.plane_v.pixels])

    @staticmethod
    def from_packed(width: int, height: int, buf: np.ndarray, padded: bool = False) -> "VideoFrame":
        if padded:
            yw, yh, cw, ch = pad16(width), pad16(height), pad16(width // 2), pad16(height // 2)
        else:
            yw, yh, cw, ch = width, height, width // 2, height // 2
        buf = np.asarray(buf, dtype=np.uint8).reshape(-1)
        assert buf.size == yw * yh + 2 * cw * ch
        o1, o2 = yw * yh, yw * yh + cw * ch
        return VideoFrame(width, height, VideoPlane.from_slice(yw, yh, buf[:o1]), VideoPlane.from_slice(cw, ch, buf[o1:o2]),
                          VideoPlane.from_slice(cw, ch, buf[o2:]))
