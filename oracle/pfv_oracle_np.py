"""numpy restatement of the pfv-rs hot path -- the SECOND, independently written oracle.

TEST INFRASTRUCTURE ONLY (see oracle/pfv_oracle.c header).  It exists so that the C oracle
is cross-checked by something written separately from the same reference lines, and to
generate the committed golden vectors under tests/golden/ (tests/golden/make_golden.py).

It is written array-at-a-time (all 8x8 subblocks of a plane at once) rather than
block-at-a-time like the C oracle, so the two share no structure.  Paths cite the
reference (relative to the reference root).  Parity status: "unpinned by upstream golden
vectors" -- the reference's tests assert nothing about DCT/quant/motion.
"""
from __future__ import annotations

import numpy as np

FP_BITS = 8  # src/dct.rs:1

# src/dct.rs:4-13 (data)
DCT_SCALE_FACTOR = np.array(
    [32, 37, 34, 26, 32, 26, 34, 37, 37, 43, 39, 31, 37, 31, 39, 43,
     34, 39, 35, 28, 34, 28, 35, 39, 26, 31, 28, 22, 26, 22, 28, 31,
     32, 37, 34, 26, 32, 26, 34, 37, 26, 31, 28, 22, 26, 22, 28, 31,
     34, 39, 35, 28, 34, 28, 35, 39, 37, 43, 39, 31, 37, 31, 39, 43], dtype=np.int32)
# src/dct.rs:16-25 (data)
Q_TABLE_INTRA = np.array(
    [8, 16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37,
     19, 22, 26, 27, 29, 34, 34, 38, 22, 22, 26, 27, 29, 34, 37, 40,
     22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32, 35, 40, 48, 58,
     26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83], dtype=np.int32)
# src/dct.rs:28-37 (data)
Q_TABLE_INTER = np.full(64, 16, dtype=np.int32)
# src/dct.rs:39-42 (data)
INV_ZIGZAG_TABLE = np.array(
    [0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42,
     3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
     10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
     21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63], dtype=np.int64)
# src/dct.rs:44-47 (data)
ZIGZAG_TABLE = np.array(
    [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
     12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
     58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


# Mutation switches, one per bit-exactness trap of SURVEY.md section 8c.  The defaults ARE the reference's rules; the only
# user of any other value is tests/test_mutation_sensitivity.py, which flips one rule at a time and asserts that the
# committed golden vectors notice (or, for the two rules that are provably unobservable, that they do not).
DEFAULT_RULES = {
    "fdct_div": "trunc",          # src/dct.rs:206-214   `/` in the forward butterfly ("floor" = arithmetic shift)
    "idct_div": "trunc",          # src/dct.rs:265-274   `/` in the inverse butterfly
    "quant_div": "trunc",         # src/dct.rs:95        n / q
    "quant_shift": "floor",       # src/dct.rs:92        (m * SCALE) >> 16 is an arithmetic shift ("trunc" = toward zero)
    "pixel_shift": "floor",       # src/common.rs:321    v >> 8
    "enc_order": "rows_cols",     # src/common.rs:294-295
    "dec_order": "cols_rows",     # src/common.rs:315-316
    "dec_table_index": "zigzag",  # src/dct.rs:78-82     decode indexes SCALE / q by zigzag position ("raster" = like encode)
    "i16_cast": "wrap",           # src/dct.rs:95        `as i16` ("saturate")
    "i32": "wrap",                # release-mode i32 arithmetic ("wide" = no wrap-around)
    "bounds": "inclusive",        # src/common.rs:171, 182   candidate allowed up to dim - 16 ("exclusive" = `>=`)
    "accept": "lt",               # src/common.rs:189    strict `<`: first visited wins ties ("le")
    "visit": "my_outer",          # src/common.rs:168-179    my outer loop, mx inner ("mx_outer")
    "skip": "le",                 # src/common.rs:221    best_err <= min_err ("lt")
    "resid_div": "trunc",         # src/common.rs:304    delta / 2 ("floor")
    "quadrants": "tl_tr_bl_br",   # src/common.rs:145-149    ("tl_bl_tr_br")
    "pad_clear": "arg",           # src/common.rs:352-356 + src/enc.rs:84-90: pad colour 0 luma / 128 chroma ("zero")
    "u8_cast": "clamp",           # src/common.rs:321, :102  clamp(0, 255) before `as u8` ("wrap" = cast without the clamp)
}
RULES = dict(DEFAULT_RULES)


def _div_pow2(x: np.ndarray, d: int, rule: str) -> np.ndarray:
    x = x.astype(np.int64)
    if rule == "floor":
        return x // d
    return (np.sign(x) * (np.abs(x) // d)).astype(np.int64)


def _tdiv(x: np.ndarray, d: int) -> np.ndarray:
    """Rust `/` on i32 by a positive power of two: truncation toward zero."""
    return _div_pow2(x, d, "trunc")


def _wrap(x: np.ndarray) -> np.ndarray:
    """wrap an int64 array to i32 two's complement (Rust release-mode arithmetic)."""
    if RULES["i32"] != "wrap":
        return np.asarray(x).astype(np.int64)
    return ((x.astype(np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int64)


def fdct(v: np.ndarray) -> np.ndarray:
    """src/dct.rs:176-239 on the LAST axis (length 8); int64 in, wrapped-i32 values out."""
    _fd = lambda x, d: _div_pow2(x, d, RULES["fdct_div"])
    i = [v[..., k].astype(np.int64) for k in range(8)]
    a0, a1, a2, a3 = _wrap(i[0] + i[7]), _wrap(i[1] + i[6]), _wrap(i[2] + i[5]), _wrap(i[3] + i[4])
    a4, a5, a6, a7 = _wrap(i[0] - i[7]), _wrap(i[1] - i[6]), _wrap(i[2] - i[5]), _wrap(i[3] - i[4])
    b0, b1, b2, b3 = _wrap(a0 + a3), _wrap(a1 + a2), _wrap(a0 - a3), _wrap(a1 - a2)
    c0, c1 = _wrap(b0 + b1), _wrap(b0 - b1)
    c2 = _wrap(_wrap(b2 + _fd(b2, 4)) + _fd(b3, 2))
    c3 = _wrap(_wrap(_fd(b2, 2) - b3) - _fd(b3, 4))
    b4 = _wrap(_wrap(_wrap(_fd(a7, 4) + a4) + _fd(a4, 4)) - _fd(a4, 16))
    b7 = _wrap(_wrap(_wrap(_fd(a4, 4) - a7) - _fd(a7, 4)) + _fd(a7, 16))
    b5 = _wrap(_wrap(_wrap(a5 + a6) - _fd(a6, 4)) - _fd(a6, 16))
    b6 = _wrap(_wrap(_wrap(a6 - a5) + _fd(a5, 4)) + _fd(a5, 16))
    c4, c5, c6, c7 = _wrap(b4 + b5), _wrap(b4 - b5), _wrap(b6 + b7), _wrap(b6 - b7)
    d4, d5, d6, d7 = c4, _wrap(c5 + c7), _wrap(c5 - c7), c6
    return np.stack([c0, d4, c2, d6, c1, d5, c3, d7], axis=-1)


def idct(v: np.ndarray) -> np.ndarray:
    """src/dct.rs:241-293 on the LAST axis."""
    _id = lambda x, d: _div_pow2(x, d, RULES["idct_div"])
    c0, d4, c2, d6, c1, d5, c3, d7 = [v[..., k].astype(np.int64) for k in range(8)]
    c4, c5, c7, c6 = d4, _wrap(d5 + d6), _wrap(d5 - d6), d7
    b4, b5, b6, b7 = _wrap(c4 + c5), _wrap(c4 - c5), _wrap(c6 + c7), _wrap(c6 - c7)
    b0, b1 = _wrap(c0 + c1), _wrap(c0 - c1)
    b2 = _wrap(_wrap(c2 + _id(c2, 4)) + _id(c3, 2))
    b3 = _wrap(_wrap(_id(c2, 2) - c3) - _id(c3, 4))
    a4 = _wrap(_wrap(_wrap(_id(b7, 4) + b4) + _id(b4, 4)) - _id(b4, 16))
    a7 = _wrap(_wrap(_wrap(_id(b4, 4) - b7) - _id(b7, 4)) + _id(b7, 16))
    a5 = _wrap(_wrap(_wrap(b5 - b6) + _id(b6, 4)) + _id(b6, 16))
    a6 = _wrap(_wrap(_wrap(b6 + b5) - _id(b5, 4)) - _id(b5, 16))
    a0, a1, a2, a3 = _wrap(b0 + b2), _wrap(b1 + b3), _wrap(b1 - b3), _wrap(b0 - b2)
    return np.stack([_wrap(a0 + a4), _wrap(a1 + a5), _wrap(a2 + a6), _wrap(a3 + a7),
                     _wrap(a3 - a7), _wrap(a2 - a6), _wrap(a1 - a5), _wrap(a0 - a4)], axis=-1)


def fdct2d(m: np.ndarray) -> np.ndarray:
    """rows then columns (src/common.rs:294-295); m: [..., 8(row), 8(col)]."""
    if RULES["enc_order"] != "rows_cols":
        return fdct(np.swapaxes(fdct(np.swapaxes(m, -1, -2)), -1, -2))
    m = fdct(m)                                    # each row (last axis)
    return np.swapaxes(fdct(np.swapaxes(m, -1, -2)), -1, -2)   # each column


def idct2d(m: np.ndarray) -> np.ndarray:
    """columns then rows (src/common.rs:315-316)."""
    if RULES["dec_order"] != "cols_rows":
        return np.swapaxes(idct(np.swapaxes(idct(m), -1, -2)), -1, -2)
    m = np.swapaxes(idct(np.swapaxes(m, -1, -2)), -1, -2)
    return idct(m)


def dct_encode(m: np.ndarray, q: np.ndarray) -> np.ndarray:
    """src/dct.rs:88-99; m: [..., 64] raster (wrapped i32 values) -> [..., 64] i16 zigzag."""
    raster = m.reshape(m.shape[:-1] + (64,)).astype(np.int64)
    n = _div_pow2(_wrap(raster * DCT_SCALE_FACTOR.astype(np.int64)), 1 << (FP_BITS * 2), RULES["quant_shift"])   # floor shift
    d = q.astype(np.int64)
    if RULES["quant_div"] == "floor":
        quo = n // d
    else:
        quo = np.sign(n) * (np.abs(n) // np.abs(d)) * np.sign(d)                 # truncating /
    if RULES["i16_cast"] != "wrap":
        return np.clip(quo[..., ZIGZAG_TABLE], -32768, 32767).astype(np.int16)
    return ((quo[..., ZIGZAG_TABLE] + 32768) % 65536 - 32768).astype(np.int16)     # `as i16` wraps


def dct_decode(src: np.ndarray, q: np.ndarray) -> np.ndarray:
    """src/dct.rs:75-86; src [..., 64] i16 zigzag -> [..., 64] raster wrapped i32 (as int64).
    SCALE and q are indexed by the zigzag position (asymmetric with encode)."""
    s = src.astype(np.int64)
    if RULES["dec_table_index"] != "zigzag":      # the "symmetric" variant: tables by raster index
        r = s[..., INV_ZIGZAG_TABLE]
        return _wrap(_wrap(r * DCT_SCALE_FACTOR.astype(np.int64)) * q.astype(np.int64))
    t = _wrap(_wrap(s * DCT_SCALE_FACTOR.astype(np.int64)) * q.astype(np.int64))   # per zigzag position
    return t[..., INV_ZIGZAG_TABLE]


def _to_subblocks(blocks16: np.ndarray) -> np.ndarray:
    """[n,16,16] -> [n,4,8,8] in quadrant order TL,TR,BL,BR (src/common.rs:145-149)."""
    n = blocks16.shape[0]
    b = blocks16.reshape(n, 2, 8, 2, 8)            # [n, qy, r, qx, c]
    if RULES["quadrants"] != "tl_tr_bl_br":
        return b.transpose(0, 3, 1, 2, 4).reshape(n, 4, 8, 8)
    return b.transpose(0, 1, 3, 2, 4).reshape(n, 4, 8, 8)


def _from_subblocks(sub: np.ndarray) -> np.ndarray:
    n = sub.shape[0]
    if RULES["quadrants"] != "tl_tr_bl_br":
        return sub.reshape(n, 2, 2, 8, 8).transpose(0, 2, 3, 1, 4).reshape(n, 16, 16)
    return sub.reshape(n, 2, 2, 8, 8).transpose(0, 1, 3, 2, 4).reshape(n, 16, 16)


def pad16(x: int) -> int:
    return x + (16 - (x % 16)) % 16               # src/common.rs:352-353


def pad_plane(px: np.ndarray, clear: int) -> np.ndarray:
    """src/common.rs:352-356."""
    h, w = px.shape
    out = np.full((pad16(h), pad16(w)), clear if RULES["pad_clear"] == "arg" else 0, dtype=np.uint8)
    out[:h, :w] = px
    return out


def _gather(img: np.ndarray) -> np.ndarray:
    """[H,W] -> [bh*bw,16,16] raster MB order (src/common.rs:364-369)."""
    H, W = img.shape
    return img.reshape(H // 16, 16, W // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 16, 16)


def _scatter(blocks: np.ndarray, bw: int, bh: int) -> np.ndarray:
    return blocks.reshape(bh, bw, 16, 16).transpose(0, 2, 1, 3).reshape(bh * 16, bw * 16)


def encode_blocks(blocks16: np.ndarray, q: np.ndarray) -> np.ndarray:
    """src/common.rs:141-152 + :287-298 over an array of macroblocks -> [n,256] i16."""
    sub = _to_subblocks(blocks16.astype(np.int64))
    m = (sub - 128) << FP_BITS
    coef = dct_encode(fdct2d(m).reshape(sub.shape[0], 4, 64), q)
    return coef.reshape(-1, 256)


def encode_blocks_delta(delta16: np.ndarray, q: np.ndarray) -> np.ndarray:
    """src/common.rs:300-311 over an array of i16 residual macroblocks -> [n,256] i16."""
    sub = _to_subblocks(delta16.astype(np.int64))
    m = _div_pow2(sub, 2, RULES["resid_div"]) << FP_BITS
    coef = dct_encode(fdct2d(m).reshape(sub.shape[0], 4, 64), q)
    return coef.reshape(-1, 256)


def decode_blocks(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """src/common.rs:238-252 + :313-325; coef [n,256] i16 -> [n,16,16] u8."""
    n = coef.shape[0]
    m = dct_decode(coef.reshape(n, 4, 64), q).reshape(n, 4, 8, 8)
    px = _div_pow2(idct2d(m), 1 << FP_BITS, RULES["pixel_shift"]) + 128
    if RULES["u8_cast"] != "clamp":
        return _from_subblocks((px & 255).astype(np.uint8))
    return _from_subblocks(np.clip(px, 0, 255).astype(np.uint8))


def encode_plane(px: np.ndarray, q: np.ndarray, clear: int):
    """src/common.rs:351-386 -> (coef [n,256] i16, bw, bh)."""
    img = pad_plane(px, clear)
    return encode_blocks(_gather(img), q), img.shape[1] // 16, img.shape[0] // 16


def decode_plane(coef: np.ndarray, bw: int, bh: int, q: np.ndarray) -> np.ndarray:
    """src/common.rs:423-446."""
    return _scatter(decode_blocks(coef, q), bw, bh)


def ssd(a: np.ndarray, b: np.ndarray) -> int:
    """full integer SSD; equals calc_error (src/common.rs:125-139) whenever the caller's
    comparison outcome matters."""
    d = a.astype(np.int64) - b.astype(np.int64)
    return int((d * d).sum())


def block_search(src: np.ndarray, ref: np.ndarray, cx: int, cy: int):
    """src/common.rs:154-204 with exact integer SSD instead of the early-exit f32 one."""
    H, W = ref.shape
    tdx = tdy = 0
    best = None
    step = 8
    while step >= 1:
        best = ssd(src, ref[cy:cy + 16, cx:cx + 16])
        bdx = bdy = 0
        lim = 16 if RULES["bounds"] == "inclusive" else 17
        if RULES["visit"] == "my_outer":
            order = [(mx, my) for my in (-1, 0, 1) for mx in (-1, 0, 1)]
        else:
            order = [(mx, my) for mx in (-1, 0, 1) for my in (-1, 0, 1)]
        for mx, my in order:
            oy = cy + my * step
            if oy < 0 or oy > H - lim:
                continue
            if mx == 0 and my == 0:
                continue
            ox = cx + mx * step
            if ox < 0 or ox > W - lim:
                continue
            e = ssd(src, ref[oy:oy + 16, ox:ox + 16])
            if e < best or (RULES["accept"] == "le" and e == best):
                best, bdx, bdy = e, mx * step, my * step
        cx += bdx
        cy += bdy
        tdx += bdx
        tdy += bdy
        step //= 2
    return tdx, tdy, best


def encode_plane_delta(px: np.ndarray, ref: np.ndarray, q: np.ndarray, px_err: float, clear: int):
    """src/common.rs:388-421 + :206-236 -> (mv [n,2] i8, has_coef [n] u8, coef [n,256] i16)."""
    img = pad_plane(px, clear)
    bw, bh = img.shape[1] // 16, img.shape[0] // 16
    blocks = _gather(img)
    n = bw * bh
    mv = np.zeros((n, 2), dtype=np.int8)
    has = np.zeros(n, dtype=np.uint8)
    coef = np.zeros((n, 256), dtype=np.int16)
    min_err = np.float32(px_err) * np.float32(px_err) * np.float32(256.0)
    for i in range(n):
        bx, by = (i % bw) * 16, (i // bw) * 16
        dx, dy, err = block_search(blocks[i], ref, bx, by)
        mv[i] = (dx, dy)
        if np.float32(err) < min_err or (RULES["skip"] == "le" and np.float32(err) == min_err):
            continue
        has[i] = 1
        prev = ref[by + dy:by + dy + 16, bx + dx:bx + dx + 16]
        delta = np.clip(blocks[i].astype(np.int64) - prev.astype(np.int64), -255, 255)
        coef[i] = encode_blocks_delta(delta[None], q)[0]
    return mv, has, coef


def decode_plane_delta(mv: np.ndarray, has: np.ndarray, coef: np.ndarray, bw: int, bh: int, q: np.ndarray,
                       ref: np.ndarray) -> np.ndarray:
    """src/common.rs:448-475 + :254-285 + :98-104."""
    n = bw * bh
    out = np.zeros((n, 16, 16), dtype=np.uint8)
    dec = decode_blocks(coef, q).astype(np.int64)
    for i in range(n):
        bx, by = (i % bw) * 16, (i // bw) * 16
        sx, sy = bx + int(mv[i, 0]), by + int(mv[i, 1])
        prev = ref[sy:sy + 16, sx:sx + 16].astype(np.int64)
        if has[i]:
            out[i] = np.clip(prev + (dec[i] - 128) * 2, 0, 255).astype(np.uint8)
        else:
            out[i] = prev.astype(np.uint8)
    return _scatter(out, bw, bh)


def qtables(quality: int):
    """src/enc.rs:40-51 in f32 -> (intra_l, intra_c, inter_l, inter_c, px_err)."""
    qs = np.float32(quality) * np.float32(0.25)
    half = np.float32(0.5)
    one = np.float32(1.0)

    def mk(base, luma):
        t = base.astype(np.float32) * qs
        if luma:
            t = t * half
        return np.maximum(t, one).astype(np.int32)

    return (mk(Q_TABLE_INTRA, True), mk(Q_TABLE_INTRA, False), mk(Q_TABLE_INTER, True), mk(Q_TABLE_INTER, False),
            float(np.float32(quality) * np.float32(1.5)))


# ------------------------------------------------------------------ colour helpers of the reference's tests
def _as_u8(x):
    """Rust `f32 as u8`: truncate toward zero, saturate"""
    return np.clip(np.trunc(x), 0, 255).astype(np.uint8)


def rgb_to_yuv420(rgb):
    """load_frame (src/lib.rs:337-359) + VideoFrame::from_planes (src/frame.rs:51-59); rgb [h, w, 3] u8 -> packed Y|U|V"""
    f = np.float32
    r, g, b = (rgb[..., k].astype(f) for k in range(3))
    y = (f(0.299) * r + f(0.587) * g) + f(0.114) * b
    u = ((f(128.0) - f(0.168736) * r) - f(0.331264) * g) + f(0.5) * b
    v = ((f(128.0) + f(0.5) * r) - f(0.418688) * g) - f(0.081312) * b
    h, w = rgb.shape[:2]
    return np.concatenate([_as_u8(y).reshape(-1), _as_u8(u)[0:h // 2 * 2:2, 0:w // 2 * 2:2].reshape(-1),
                           _as_u8(v)[0:h // 2 * 2:2, 0:w // 2 * 2:2].reshape(-1)])


def yuv420_to_rgb(frame, w, h):
    """save_frame (src/lib.rs:361-394): chroma doubled (nearest), JPEG-conversion YCbCr -> RGB in f32"""
    f = np.float32
    n, cw, ch = w * h, w // 2, h // 2
    y = frame[:n].reshape(h, w).astype(f)
    up = lambda p: np.repeat(np.repeat(p.reshape(ch, cw), 2, axis=0), 2, axis=1)
    u = up(frame[n:n + cw * ch]).astype(f) - f(128.0)
    v = up(frame[n + cw * ch:]).astype(f) - f(128.0)
    r = y + f(1.402) * v
    g = (y - f(0.344136) * u) - f(0.714136) * v
    b = y + f(1.772) * u
    return np.stack([_as_u8(r), _as_u8(g), _as_u8(b)], axis=-1)
