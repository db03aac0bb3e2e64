"""Deterministic, integer-only synthetic YUV 4:2:0 streams.

The reference's fixtures are Git-LFS stubs, so every test and benchmark runs on synthetic
video: a smooth random texture (coarse u8 grid, integer-bilinear upsampled 8x, plus fine
noise) that translates by ``((3t mod 23) - 11, (2t mod 17) - 8)`` luma pixels at frame t,
with +-16 per-pixel noise on about half of the macroblocks (RMS 9.5, above the quality-5 skip
threshold of 7.5 per pixel, src/enc.rs:41 + src/common.rs:209) so that both skipped and coded
p-frame macroblocks occur in roughly equal numbers.  Everything is a counter-based 64-bit integer hash (splitmix64
finaliser) evaluated with numpy uint64 arithmetic -- no floats, no library RNG state -- so
the same bytes come out on every machine.
"""
from __future__ import annotations

import numpy as np

SEED = 0x50465632  # "PFV2"
_MARGIN = 32


def _hash64(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _texture(h: int, w: int, seed: int) -> np.ndarray:
    """smooth texture of shape (h, w), int32 values in 0..255"""
    gh, gw = h // 8 + 2, w // 8 + 2
    grid = (_hash64(np.arange(gh * gw), seed) & np.uint64(0xFF)).astype(np.int32).reshape(gh, gw)
    y, x = np.arange(h), np.arange(w)
    gy, fy = (y >> 3)[:, None], (y & 7)[:, None]
    gx, fx = (x >> 3)[None, :], (x & 7)[None, :]
    top = (8 - fx) * grid[gy, gx] + fx * grid[gy, gx + 1]
    bot = (8 - fx) * grid[gy + 1, gx] + fx * grid[gy + 1, gx + 1]
    tex = ((8 - fy) * top + fy * bot) >> 6
    fine = (_hash64(np.arange(h * w), seed ^ 0x5EED) % np.uint64(7)).astype(np.int32).reshape(h, w) - 3
    return np.clip(tex + fine, 0, 255)


class SyntheticStream:
    def __init__(self, width: int, height: int, seed: int = SEED):
        assert width % 2 == 0 and height % 2 == 0
        self.width, self.height, self.seed = width, height, seed
        self._dims = [(height, width), (height // 2, width // 2), (height // 2, width // 2)]
        self._tex = [_texture(h + 2 * _MARGIN, w + 2 * _MARGIN, seed + 101 * p) for p, (h, w) in enumerate(self._dims)]

    @staticmethod
    def motion(t: int):
        return (3 * t) % 23 - 11, (2 * t) % 17 - 8

    def plane(self, t: int, p: int) -> np.ndarray:
        h, w = self._dims[p]
        ox, oy = self.motion(t)
        if p:
            ox, oy = ox // 2, oy // 2
        img = self._tex[p][_MARGIN + oy:_MARGIN + oy + h, _MARGIN + ox:_MARGIN + ox + w].copy()
        bw, bh = (w + 15) // 16, (h + 15) // 16
        noisy = (_hash64(np.arange(bw * bh), self.seed + 7919 * t + p) & np.uint64(1)).astype(bool).reshape(bh, bw)
        mask = np.repeat(np.repeat(noisy, 16, axis=0), 16, axis=1)[:h, :w]
        noise = (_hash64(np.arange(h * w), self.seed + 104729 * t + 13 * p) % np.uint64(33)).astype(np.int32).reshape(h, w) - 16
        img = np.where(mask, img + noise, img)
        return np.clip(img, 0, 255).astype(np.uint8)

    def frame(self, t: int) -> np.ndarray:
        """packed Y|U|V frame t (uint8, pfv_frame_bytes long)"""
        return np.concatenate([self.plane(t, p).reshape(-1) for p in range(3)])
