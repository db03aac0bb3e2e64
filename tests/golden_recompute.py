"""Recomputes every committed golden output (tests/golden/hotpath_vectors.npz, trap_vectors.npz) from its committed INPUTS
with the numpy oracle.  Used by tests/golden/make_trap_vectors.py (to produce the trap vectors), by tests/test_oracle.py
(unmutated oracle == committed goldens) and by tests/test_mutation_sensitivity.py (one rule flipped -> something changes)."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import pfv_oracle_np as onp   # noqa: E402

GOLD = os.path.join(HERE, "golden", "hotpath_vectors.npz")
TRAP = os.path.join(HERE, "golden", "trap_vectors.npz")


def _subblocks_as_mbs(sub64):
    """[n, 64] subblocks (n % 4 == 0) -> [n/4, 16, 16] macroblocks holding them as TL,TR,BL,BR (reference order)"""
    s = np.asarray(sub64).reshape(-1, 2, 2, 8, 8)
    return s.transpose(0, 1, 3, 2, 4).reshape(-1, 16, 16)


def _mbs_as_subblocks(mb):
    n = mb.shape[0]
    return mb.reshape(n, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(n * 4, 64)


def recompute_hotpath(g):
    """outputs of hotpath_vectors.npz from its inputs (ramp, lib_block, lib_q, sub_px, sub_delta, ext_delta, pf_f0, pf_f1)"""
    out = {}
    out["ramp_fdct"] = onp.fdct(g["ramp"][None].astype(np.int64))[0].astype(np.int32)
    out["ramp_idct"] = onp.idct(g["ramp_fdct"][None].astype(np.int64))[0].astype(np.int32)
    m = (g["lib_block"].astype(np.int64).reshape(8, 8) - 128) << 8
    pre = onp.fdct2d(m[None])[0].reshape(64)
    out["lib_prequant"] = pre.astype(np.int32)
    out["lib_quant"] = onp.dct_encode(pre[None], g["lib_q"])[0]
    out["lib_recon"] = onp.decode_blocks(np.tile(g["lib_quant"], 4)[None], g["lib_q"])[0][:8, :8].reshape(64)
    px_mb = _subblocks_as_mbs(g["sub_px"])
    dl_mb = _subblocks_as_mbs(g["sub_delta"])
    for quality in (0, 2, 5, 10):
        tabs = onp.qtables(quality)
        for name, q in (("intra_l", tabs[0]), ("intra_c", tabs[1]), ("inter_l", tabs[2])):
            out[f"q{quality}_{name}_enc"] = onp.encode_blocks(px_mb, q).reshape(-1, 64)
            out[f"q{quality}_{name}_encdelta"] = onp.encode_blocks_delta(dl_mb, q).reshape(-1, 64)
            out[f"q{quality}_{name}_dec"] = _mbs_as_subblocks(onp.decode_blocks(g[f"q{quality}_{name}_enc"].reshape(-1, 256), q))
    ext_mb = _subblocks_as_mbs(np.tile(g["ext_delta"], (4, 1)))
    out["ext_out"] = np.stack([onp.encode_blocks_delta(ext_mb, q)[0, :64] for q in (onp.qtables(1)[2], onp.qtables(10)[3])])
    il, _, pl, _, px_err = onp.qtables(5)
    c0, bw, bh = onp.encode_plane(g["pf_f0"], il, 0)
    out["pf_c0"] = c0
    out["pf_rec0"] = onp.decode_plane(g["pf_c0"], bw, bh, il)
    mv, has, c1 = onp.encode_plane_delta(g["pf_f1"], g["pf_rec0"], pl, px_err, 0)
    out["pf_mv"], out["pf_has"], out["pf_c1"] = mv, has, c1
    out["pf_rec1"] = onp.decode_plane_delta(g["pf_mv"], g["pf_has"], g["pf_c1"], bw, bh, pl, g["pf_rec0"])
    return out


def recompute_traps(t):
    """outputs of trap_vectors.npz from its inputs (rag_*, tie_*, diag_*, host_*)"""
    out = {}
    _, ic, _, pc, px_err = onp.qtables(5)
    # ragged chroma-like plane: padding colour matters (src/common.rs:352-356, src/enc.rs:84-90)
    c, bw, bh = onp.encode_plane(t["rag_f0"], ic, 128)
    out["rag_c0"] = c
    out["rag_rec0"] = onp.decode_plane(t["rag_c0"], bw, bh, ic)
    mv, has, c1 = onp.encode_plane_delta(t["rag_f1"], t["rag_rec0"], pc, px_err, 128)
    out["rag_mv"], out["rag_has"], out["rag_c1"] = mv, has, c1
    out["rag_rec1"] = onp.decode_plane_delta(t["rag_mv"], t["rag_has"], t["rag_c1"], bw, bh, pc, t["rag_rec0"])
    # exact ties + skip threshold met with equality (src/common.rs:189, :221)
    pl = onp.qtables(5)[2]
    mv, has, c1 = onp.encode_plane_delta(t["tie_src"], t["tie_ref"], pl, px_err, 0)
    out["tie_mv"], out["tie_has"], out["tie_c1"] = mv, has, c1
    # diagonal stripes: candidates with equal dx + dy tie, the visiting order decides (src/common.rs:168-179)
    mv, has, c1 = onp.encode_plane_delta(t["diag_src"], t["diag_ref"], pl, px_err, 0)
    out["diag_mv"], out["diag_has"], out["diag_c1"] = mv, has, c1
    # a best candidate that sits exactly on the plane's last legal position (src/common.rs:171, :182)
    mv, has, c1 = onp.encode_plane_delta(t["edge_src"], t["edge_ref"], pl, px_err, 0)
    out["edge_mv"], out["edge_has"], out["edge_c1"] = mv, has, c1
    # hostile coefficients and a q-table up to 65535: i32 wrap-around in decode (src/dct.rs:75-86, release arithmetic)
    out["host_rec"] = onp.decode_blocks(t["host_coef"], t["host_q"])
    return out


def load():
    return np.load(GOLD), np.load(TRAP)


def diff_keys(want, got):
    return sorted(k for k in got if not np.array_equal(np.asarray(want[k]), np.asarray(got[k])))
