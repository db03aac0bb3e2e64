"""Mutation sensitivity of the committed golden vectors ("missing" #1).

Nothing upstream can pin the oracle (the reference's DCT tests only print, src/lib.rs:36-94; its fixtures are Git-LFS stubs; no
Rust toolchain), so the goldens under tests/golden/ are oracle output.  What CAN be shown is that they are not blind: for
every bit-exactness trap of SURVEY.md section 8c, flipping that one rule in the numpy oracle (oracle/pfv_oracle_np.py RULES)
changes at least one committed vector -- so a HIP kernel (or a C oracle) that got the rule wrong could not pass
tests/test_gpu_parity.py::test_golden_vectors / test_trap_vectors.  Two rules are unobservable BY THEOREM, and the test
asserts exactly that, with the argument.
"""
import numpy as np
import pytest

import golden_recompute as gr
from golden_recompute import onp

# (rule, mutated value, reference lines, vectors that must notice -- a non-empty subset is required)
VISIBLE = [
    ("idct_div", "floor", "src/dct.rs:265-274 `/` truncates toward zero", ("ramp_idct", "lib_recon", "pf_rec0")),
    ("quant_div", "floor", "src/dct.rs:95 n / q truncates", ("lib_quant", "pf_c0")),
    ("quant_shift", "trunc", "src/dct.rs:92 >> 16 is an arithmetic shift (floor)", ("lib_quant", "pf_c0")),
    ("pixel_shift", "trunc", "src/common.rs:321 >> 8 is an arithmetic shift", ("lib_recon", "pf_rec0")),
    ("dec_order", "rows_cols", "src/common.rs:315-316 decode = columns then rows", ("lib_recon", "pf_rec0")),
    ("dec_table_index", "raster", "src/dct.rs:78-82 decode tables indexed by zigzag position", ("lib_recon", "pf_rec0")),
    ("i32", "wide", "release-mode i32 wrap-around (SURVEY 8c trap 11)", ("host_rec",)),
    ("u8_cast", "wrap", "src/common.rs:321 clamp before `as u8`", ("host_rec", "q10_intra_c_dec")),
    ("bounds", "exclusive", "src/common.rs:171, :182 `>` (candidate allowed at dim - 16)", ("edge_mv",)),
    ("accept", "le", "src/common.rs:189 strict `<`: first visited wins ties", ("tie_mv",)),
    ("visit", "mx_outer", "src/common.rs:168-179 my outer, mx inner", ("diag_mv",)),
    ("skip", "lt", "src/common.rs:221 `<=` against 576 q^2 (:209)", ("tie_has",)),
    ("resid_div", "floor", "src/common.rs:304 delta / 2 truncates", ("ext_out", "pf_c1")),
    ("quadrants", "tl_bl_tr_br", "src/common.rs:145-149 subblocks TL,TR,BL,BR", ("pf_c0", "pf_rec0")),
    ("pad_clear", "zero", "src/common.rs:352-356 + src/enc.rs:84-90 pad colour 128 for chroma", ("rag_c0",)),
]

# Unobservable by theorem -- asserted to change NOTHING, here and (for the forward transform) exhaustively in
# tests/test_oracle.py::test_forward_dct_is_exact:
INVISIBLE = [
    ("fdct_div", "floor",
     "src/dct.rs:206-214: every fdct input has 8 zero fraction bits (common.rs:291, :304), so the row pass divides multiples of 256 and "
     "the column pass multiples of 16: no division of the forward transform ever truncates"),
    ("enc_order", "cols_rows",
     "src/common.rs:294-295: for the same reason both forward passes are exact linear maps on different axes; they commute"),
    ("i16_cast", "saturate",
     "src/dct.rs:95: n = (i32) >> 16 lies in [-32768, 32767] and |n / q| <= |n| for q >= 1, so `as i16` never wraps"),
]


@pytest.fixture(scope="module")
def vectors():
    g, t = gr.load()
    return {k: g[k] for k in g.files}, {k: t[k] for k in t.files}


def _recompute(g, t):
    out = gr.recompute_hotpath(g)
    out.update(gr.recompute_traps(t))
    return out


def test_unmutated_numpy_oracle_reproduces_every_golden(vectors):
    g, t = vectors
    assert onp.RULES == onp.DEFAULT_RULES
    got = _recompute(g, t)
    want = {**g, **t}
    assert gr.diff_keys(want, got) == []
    assert len(got) >= 60


@pytest.mark.parametrize("rule,value,where,must", VISIBLE, ids=[v[0] for v in VISIBLE])
def test_golden_vectors_notice_a_flipped_rule(vectors, rule, value, where, must):
    g, t = vectors
    want = {**g, **t}
    onp.RULES[rule] = value
    try:
        changed = gr.diff_keys(want, _recompute(g, t))
    finally:
        onp.RULES.update(onp.DEFAULT_RULES)
    assert changed, f"no committed vector notices {rule}={value} ({where})"
    assert set(must) & set(changed), f"{rule}={value}: expected one of {must} to change, got {changed}"


@pytest.mark.parametrize("rule,value,why", INVISIBLE, ids=[v[0] for v in INVISIBLE])
def test_rules_unobservable_by_theorem(vectors, rule, value, why):
    g, t = vectors
    want = {**g, **t}
    onp.RULES[rule] = value
    try:
        changed = gr.diff_keys(want, _recompute(g, t))
    finally:
        onp.RULES.update(onp.DEFAULT_RULES)
    assert changed == [], f"{rule}={value} was expected to be unobservable ({why}) but changed {changed}"


def test_every_rule_is_covered():
    assert {v[0] for v in VISIBLE} | {v[0] for v in INVISIBLE} == set(onp.DEFAULT_RULES)
