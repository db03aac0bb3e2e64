#!/usr/bin/env python3
"""Generates tests/golden/trap_vectors.npz: known-answer vectors aimed at the bit-exactness traps of SURVEY.md section 8c that
tests/golden/hotpath_vectors.npz does not exercise (found by tests/test_mutation_sensitivity.py: flipping the rule left
every older vector unchanged): pad colour on a ragged plane, exact ties in the motion search, the skip threshold met with
equality, a best match on the last legal position of the plane, i32 wrap-around in decode.

Like the older vectors they are produced by the numpy oracle and cross-checked against the C oracle before being written
(the Rust reference cannot run here and holds no expected values for this path).  python tests/golden/make_trap_vectors.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pfv_oracle_np as onp          # noqa: E402
import golden_recompute as gr        # noqa: E402
from oracle_bind import Oracle        # noqa: E402

ora = Oracle()


def smooth(h, w, seed, amp=3):
    r = np.random.default_rng(seed)
    g = r.integers(0, 256, (h // 8 + 2, w // 8 + 2)).astype(np.int32)
    y, x = np.arange(h), np.arange(w)
    gy, fy, gx, fx = (y >> 3)[:, None], (y & 7)[:, None], (x >> 3)[None, :], (x & 7)[None, :]
    t = ((8 - fy) * ((8 - fx) * g[gy, gx] + fx * g[gy, gx + 1]) + fy * ((8 - fx) * g[gy + 1, gx] + fx * g[gy + 1, gx + 1])) >> 6
    return np.clip(t + r.integers(-amp, amp + 1, (h, w)), 0, 255).astype(np.uint8)


inp = {}
# ---- ragged 50 x 38 "chroma" plane (pads to 64 x 48 with 128)
big = smooth(38 + 32, 50 + 32, 21)
inp["rag_f0"] = np.clip(big[16:54, 16:66].astype(int) - 60, 0, 255).astype(np.uint8)     # dark content: the pad colour is far from it
inp["rag_f1"] = np.clip(big[16 + 2:54 + 2, 16 - 3:66 - 3].astype(int) - 60 + np.random.default_rng(22).integers(-12, 13, (38, 50)), 0, 255).astype(np.uint8)

# ---- ties and the skip threshold: rows 0..31 of the reference are flat (100), rows 32..47 textured
ref = np.full((48, 64), 100, np.uint8)
ref[32:] = smooth(16, 64, 23, amp=20)
src = ref.copy()
blk = np.full(256, 100, np.uint8); blk[:144] = 110                      # SSD 144 * 100 = 14 400 = 576 * 5^2 exactly
src[0:16, 16:32] = blk.reshape(16, 16)
blk = np.full(256, 100, np.uint8); blk[:143] = 110; blk[143] = 111      # 14 421: just above
src[0:16, 32:48] = blk.reshape(16, 16)
blk = np.full(256, 100, np.uint8); blk[:143] = 110; blk[143] = 109      # 14 381: just below
src[0:16, 48:64] = blk.reshape(16, 16)
src[32:] = np.clip(ref[32:].astype(int) + np.random.default_rng(24).integers(-30, 31, (16, 64)), 0, 255)
inp["tie_ref"], inp["tie_src"] = ref, src

# ---- diagonal stripes f(x + y): all displacements with the same dx + dy are indistinguishable
g1 = np.random.default_rng(25).integers(0, 256, 200).astype(np.uint8)
yy, xx = np.mgrid[0:64, 0:80]
dref = g1[xx + yy]
inp["diag_ref"] = dref
inp["diag_src"] = g1[np.clip(xx + yy + 9, 0, 199)]                       # the content moved by dx + dy = 9

# ---- best match on the last legal row / column: content displaced so that the match sits at x = W - 16, y = H - 16
eref = smooth(48, 64, 26, amp=25)
esrc = eref.copy()
esrc[16:32, 32:48] = eref[32:48, 48:64]                                   # MB (2,1) matches the bottom-right corner block: (+16,+16) is out of reach,
esrc[16:32, 16:32] = eref[31:47, 16:32]                                   # (0,+15) is the last legal row... for the macroblock row above it
esrc[0:16, 32:48] = eref[0:16, 47:63]                                     # (+15, 0)
esrc[32:48, 48:64] = eref[32:48, 33:49]                                   # bottom-right macroblock looking left: (-15, 0); +x, +y candidates illegal
inp["edge_ref"], inp["edge_src"] = eref, esrc

# ---- hostile decode input
r = np.random.default_rng(27)
inp["host_coef"] = r.integers(-32768, 32768, (8, 256)).astype(np.int16)
inp["host_q"] = r.integers(1, 65536, 64).astype(np.int32)

# outputs: numpy oracle; inputs that are themselves outputs of an earlier stage are filled in stage order
t = dict(inp)
_, ic, _, pc, px_err = onp.qtables(5)
c, bw, bh = onp.encode_plane(t["rag_f0"], ic, 128)
t["rag_c0"] = c
t["rag_rec0"] = onp.decode_plane(c, bw, bh, ic)
mv, has, c1 = onp.encode_plane_delta(t["rag_f1"], t["rag_rec0"], pc, px_err, 128)
t["rag_mv"], t["rag_has"], t["rag_c1"] = mv, has, c1
out = gr.recompute_traps(t)
t.update(out)

# ---- cross-check against the C oracle
pl = onp.qtables(5)[2]
assert np.array_equal(t["rag_c0"], ora.encode_plane(t["rag_f0"], ic, 128)[0])
assert np.array_equal(t["rag_rec0"], ora.decode_plane(t["rag_c0"], bw, bh, ic))
for name, q, clear in (("rag", pc, 128),):
    omv, ohas, oc = ora.encode_plane_delta(t["rag_f1"], t["rag_rec0"], q, px_err, clear)
    assert np.array_equal(omv, t["rag_mv"]) and np.array_equal(ohas, t["rag_has"]) and np.array_equal(oc, t["rag_c1"])
    assert np.array_equal(t["rag_rec1"], ora.decode_plane_delta(t["rag_mv"], t["rag_has"], t["rag_c1"], bw, bh, q, t["rag_rec0"]))
for name in ("tie", "diag", "edge"):
    omv, ohas, oc = ora.encode_plane_delta(t[f"{name}_src"], t[f"{name}_ref"], pl, px_err, 0)
    assert np.array_equal(omv, t[f"{name}_mv"]), name
    assert np.array_equal(ohas, t[f"{name}_has"]), name
    assert np.array_equal(oc, t[f"{name}_c1"]), name
hq = t["host_q"]
assert np.array_equal(t["host_rec"], np.stack([np.stack([ora.decode_subblock(t["host_coef"][i, k * 64:(k + 1) * 64], hq) for k in range(4)])
                                               for i in range(8)]).reshape(8, 2, 2, 8, 8).transpose(0, 1, 3, 2, 4).reshape(8, 16, 16))
np.savez_compressed(os.path.join(HERE, "trap_vectors.npz"), **t)
print("wrote trap_vectors.npz;", "tie mv", t["tie_mv"].tolist(), "has", t["tie_has"].tolist())
print("diag mv", t["diag_mv"].tolist())
print("edge mv", t["edge_mv"].tolist())
