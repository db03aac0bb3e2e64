"""Deterministic, integer-only synthetic YUV 4:2:0 streams.

The reference's fixtures are Git-LFS stubs, so every test and benchmark runs on synthetic
video: a smooth random texture (coarse u8 grid, integer-bilinear upsampled 8x, plus fine
noise) that translates by ``((3t mod 23) - 11, (2t mod 17) - 8)`` luma pixels at frame t,
with +-16 per-pixel noise on about half of the macroblocks (RMS 9.5, above the quality-5 skip
threshold of 7.5 per pixel, src/enc.rs:41 + src/common.rs:209) so that both skipped and coded
p-frame macroblocks occur in roughly equal numbers.  Everything is a counter-based 64-bit integer hash (splitmix64
finaliser) evaluated with numpy uint64 arithmetic -- no floats, no library RNG state -- so
the same bytes come out on every machine.

``kind="low_motion"``: the texture stays put (a static background that a p-frame skips, src/common.rs:221-222) and four
rectangles of about a quarter of the frame's width and height -- own texture, +-16 noise on every pixel, a few pixels of
motion per frame, edges not aligned to macroblocks -- move over it: about a quarter of a quality-5 p-frame is coded.
``kind="static"``: the background alone -- every p-frame macroblock is skipped (the floor of the p-frame encoder: its search).
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


N_OBJECTS = 4


class SyntheticStream:
    def __init__(self, width: int, height: int, seed: int = SEED, kind: str = "pan"):
        assert width % 2 == 0 and height % 2 == 0 and kind in ("pan", "low_motion", "static")
        self.width, self.height, self.seed, self.kind = width, height, seed, kind
        self._dims = [(height, width), (height // 2, width // 2), (height // 2, width // 2)]
        self._tex = [_texture(h + 2 * _MARGIN, w + 2 * _MARGIN, seed + 101 * p) for p, (h, w) in enumerate(self._dims)]
        self._obj_tex = {}

    def _object(self, j: int, t: int):
        """(x, y, w, h) of object j at frame t in luma pixels (csrc/pfv_synth_kernels.hip: synth_object)"""
        W, H = self.width, self.height
        hsh = int(_hash64(np.array([j]), self.seed + 31337)[0])
        ow = max(2, min(W, (W // 4 + (hsh & 0xFFFF) % (W // 16 + 1)) & ~1))
        oh = max(2, min(H, (H // 4 + ((hsh >> 16) & 0xFFFF) % (H // 16 + 1)) & ~1))
        rx, ry = W - ow + 1, H - oh + 1
        x0, y0 = ((hsh >> 32) & 0xFFFF) % rx, ((hsh >> 48) & 0xFFFF) % ry
        h2 = int(_hash64(np.array([j]), self.seed + 424243)[0])
        vx, vy = (h2 & 0xFF) % 13 - 6, ((h2 >> 8) & 0xFF) % 9 - 4
        return (x0 + vx * t) % rx, (y0 + vy * t) % ry, ow, oh

    def _plane_low_motion(self, t: int, p: int) -> np.ndarray:
        h, w = self._dims[p]
        img = self._tex[p][_MARGIN:_MARGIN + h, _MARGIN:_MARGIN + w].copy()
        for j in range(N_OBJECTS if self.kind == "low_motion" else 0):      # "static": the background alone
            ox, oy, ow, oh = self._object(j, t)
            if p:
                ox, oy, ow, oh = ox >> 1, oy >> 1, ow >> 1, oh >> 1
            if ow == 0 or oh == 0:
                continue
            key = (p, j)
            if key not in self._obj_tex:
                self._obj_tex[key] = _texture(oh, ow, self.seed + 101 * p + 1009 * (j + 1))
            noise = (_hash64(np.arange(oh * ow), self.seed + 104729 * t + 13 * p + 977 * (j + 1)) % np.uint64(33)).astype(np.int32).reshape(oh, ow) - 16
            img[oy:oy + oh, ox:ox + ow] = np.clip(self._obj_tex[key] + noise, 0, 255)
        return img.astype(np.uint8)

    @staticmethod
    def motion(t: int):
        return (3 * t) % 23 - 11, (2 * t) % 17 - 8

    def plane(self, t: int, p: int) -> np.ndarray:
        if self.kind != "pan":
            return self._plane_low_motion(t, p)
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
