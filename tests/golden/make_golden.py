#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- known-answer vectors for the hot path.

The reference (Rust) cannot be run in this image and ships no golden outputs for this path (its DCT tests
only print, src/lib.rs:36-94; all binary fixtures are Git-LFS stubs), so the vectors are produced by the
numpy oracle (oracle/pfv_oracle_np.py) and cross-checked against the C oracle before being written; the
two INPUTS taken from the reference's own tests are the ramp of src/lib.rs:38 and the 8x8 block + q-table
of src/lib.rs:61-66.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pfv_oracle_np as onp          # noqa: E402
from oracle_bind import Oracle        # noqa: E402

ora = Oracle()

# ---- (1) src/lib.rs:38 ramp through the 1-D transforms
ramp = (np.arange(8) * 10 << 8).astype(np.int32)
ramp_f = onp.fdct(ramp[None])[0].astype(np.int32)
assert np.array_equal(ramp_f, ora.fdct8(ramp))
ramp_i = onp.idct(ramp_f[None])[0].astype(np.int32)
assert np.array_equal(ramp_i, ora.idct8(ramp_f))

# ---- (2) src/lib.rs:61-66 block, q-table = quality-5 luma intra table
q_lib = np.array([5, 10, 11, 13, 16, 16, 18, 21, 10, 10, 13, 15, 16, 18, 21, 23, 11, 13, 16, 16, 18, 21, 21, 23, 13, 13, 16,
                  16, 18, 21, 23, 25, 13, 16, 16, 18, 20, 21, 25, 30, 16, 16, 18, 20, 21, 25, 30, 36, 16, 16, 18, 21, 23, 28,
                  35, 43, 16, 18, 21, 23, 28, 35, 43, 51], dtype=np.int32)
blk = np.array([44, 42, 43, 43, 46, 49, 42, 33, 36, 49, 56, 47, 42, 41, 36, 28, 36, 48, 57, 52, 42, 35, 29, 23, 36, 35, 41,
                48, 45, 32, 25, 24, 32, 27, 30, 39, 41, 32, 25, 26, 26, 27, 29, 30, 31, 31, 27, 23, 29, 27, 27, 27, 30, 31,
                26, 20, 35, 23, 19, 27, 34, 30, 22, 16], dtype=np.uint8)
m = ((blk.astype(np.int64).reshape(8, 8) - 128) << 8)
blk_pre = onp.fdct2d(m[None])[0].reshape(64).astype(np.int32)
blk_q = onp.dct_encode(blk_pre[None].astype(np.int64), q_lib)[0]
assert np.array_equal(blk_q, ora.encode_subblock(blk, q_lib))
blk_rec = ora.decode_subblock(blk_q, q_lib)
assert np.array_equal(blk_rec.reshape(8, 8), onp.decode_blocks(np.tile(blk_q, 4)[None], q_lib)[0][:8, :8])

# ---- (3) random subblocks x qualities x {intra_l, intra_c, inter}
rng = np.random.default_rng(2024)
sub_px = rng.integers(0, 256, (64, 64), dtype=np.uint8)
sub_delta = rng.integers(-255, 256, (64, 64)).astype(np.int16)
sub_out = {}
for quality in (0, 2, 5, 10):
    il, ic, pl, pc, _ = onp.qtables(quality)
    for name, q in (("intra_l", il), ("intra_c", ic), ("inter_l", pl)):
        enc = np.stack([ora.encode_subblock(b, q) for b in sub_px])
        encd = np.stack([ora.encode_subblock_delta(d, q) for d in sub_delta])
        dec = np.stack([ora.decode_subblock(c, q) for c in enc])
        # numpy cross-check (as 16x16 macroblocks made of 4 copies)
        for k in range(0, 64, 16):
            mb = np.zeros((16, 16), np.uint8); mb[:8, :8] = sub_px[k].reshape(8, 8)
            assert np.array_equal(onp.encode_blocks(mb[None], q)[0][:64], enc[k])
        sub_out[f"q{quality}_{name}_enc"] = enc
        sub_out[f"q{quality}_{name}_encdelta"] = encd
        sub_out[f"q{quality}_{name}_dec"] = dec

# ---- (4) 64x48 two-frame p-frame case with known translation incl. border macroblocks
def smooth(h, w, seed):
    r = np.random.default_rng(seed)
    g = r.integers(0, 256, (h // 8 + 2, w // 8 + 2)).astype(np.int32)
    y, x = np.arange(h), np.arange(w)
    gy, fy, gx, fx = (y >> 3)[:, None], (y & 7)[:, None], (x >> 3)[None, :], (x & 7)[None, :]
    t = ((8 - fy) * ((8 - fx) * g[gy, gx] + fx * g[gy, gx + 1]) + fy * ((8 - fx) * g[gy + 1, gx] + fx * g[gy + 1, gx + 1])) >> 6
    return np.clip(t + r.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)

big = smooth(48 + 32, 64 + 32, 7)
f0 = big[16:64, 16:80].copy()
f1 = big[16 - 3:64 - 3, 16 + 5:80 + 5].copy()       # translated by (+5, -3)
f1[16:32, 32:48] = np.clip(f1[16:32, 32:48].astype(int) + np.random.default_rng(9).integers(-40, 41, (16, 16)), 0, 255)
il, ic, pl, pc, px_err = onp.qtables(5)
c0, bw, bh = onp.encode_plane(f0, il, 0)
rec0 = onp.decode_plane(c0, bw, bh, il)
mv, has, c1 = onp.encode_plane_delta(f1, rec0, pl, px_err, 0)
rec1 = onp.decode_plane_delta(mv, has, c1, bw, bh, pl, rec0)
omv, ohas, oc1 = ora.encode_plane_delta(f1, rec0, pl, px_err, 0)
assert np.array_equal(mv, omv) and np.array_equal(has, ohas) and np.array_equal(c1, oc1)
assert np.array_equal(rec1, ora.decode_plane_delta(mv, has, c1, bw, bh, pl, rec0))
assert np.array_equal(c0, ora.encode_plane(f0, il, 0)[0])

# ---- (5) residual extremes (+-255)
ext = np.array([255, -255] * 32, dtype=np.int16)
ext_out = np.stack([ora.encode_subblock_delta(ext, q) for q in (onp.qtables(1)[2], onp.qtables(10)[3])])

np.savez_compressed(os.path.join(HERE, "hotpath_vectors.npz"),
                    ramp=ramp, ramp_fdct=ramp_f, ramp_idct=ramp_i,
                    lib_block=blk, lib_q=q_lib, lib_prequant=blk_pre, lib_quant=blk_q, lib_recon=blk_rec,
                    sub_px=sub_px, sub_delta=sub_delta, ext_delta=ext, ext_out=ext_out,
                    pf_f0=f0, pf_f1=f1, pf_c0=c0, pf_rec0=rec0, pf_mv=mv, pf_has=has, pf_c1=c1, pf_rec1=rec1,
                    **sub_out)
print("wrote", os.path.join(HERE, "hotpath_vectors.npz"))
print("ramp fdct", ramp_f.tolist(), "| lib block quant head", blk_q[:12].tolist(), "| pre-quant DC", int(blk_pre[0]))
print("p-frame case: mv", mv.tolist(), "has", has.tolist())
