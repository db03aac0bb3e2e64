"""Host entropy layer, CPU only: the product's packet serialisers (C++, in libpfv_hip.so, no
GPU needed for these entry points) against the C oracle's restatement of src/rle.rs / src/huffman.rs /
src/enc.rs:237-481, plus the reference's own entropy test vector (src/lib.rs:98)."""
import ctypes
import os

import numpy as np
import pytest


def _oracle_payloads(oracle):
    L = oracle.L
    L.pfvo_serialize_iframe.restype = ctypes.c_size_t
    L.pfvo_serialize_iframe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    L.pfvo_serialize_pframe.restype = ctypes.c_size_t
    L.pfvo_serialize_pframe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    L.pfvo_entropy_roundtrip.restype = ctypes.c_size_t
    L.pfvo_entropy_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_reference_entropy_vector_roundtrip(oracle):
    """src/lib.rs:96-158 test_entropy: [10,0,0,5,3,0,0,0,0,-10] survives RLE -> Huffman -> LE bits -> back"""
    L = _oracle_payloads(oracle)
    d = np.array([10, 0, 0, 5, 3, 0, 0, 0, 0, -10], np.int16)
    coded, dec, tab = np.zeros(64, np.uint8), np.zeros(10, np.int16), np.zeros(16, np.uint8)
    n = L.pfvo_entropy_roundtrip(_p(d), 10, _p(coded), 64, _p(dec), _p(tab))
    assert np.array_equal(dec, d)
    assert n == 5 and coded[:5].tolist() == [171, 88, 141, 165, 5]      # regression pin of the restated bit layout
    # symbols used: runs {0,2,4}, sizes {3,4,5}; histogram normalised to max 255 (rle.rs:49-63)
    assert tab.tolist()[:6] == [255, 0, 127, 127, 255, 255]


@pytest.mark.parametrize("density", [0.0, 0.02, 0.3, 1.0])
def test_payload_serialisers_product_equals_oracle(graft, pkg, oracle, density):
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    lib = pkg._lib.load()
    L = _oracle_payloads(oracle)
    rng = np.random.default_rng(int(density * 100))
    nb = 37
    coef = (rng.integers(-16383, 16384, (nb, 256)) * (rng.random((nb, 256)) < density)).astype(np.int16)
    coef[3] = 0                                           # an all-zero macroblock: 17 x (15,0) + (1,0)
    coef[5, 255] = -1                                     # long zero run before the last coefficient
    cap = nb * 256 * 4 + 64
    a, b = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
    na = lib.pfv_serialize_iframe_payload(_p(coef), nb, _p(a), cap)
    nbo = L.pfvo_serialize_iframe(_p(coef), nb, _p(b), cap)
    assert na == nbo > 19 and np.array_equal(a[:na], b[:na])
    assert a[16:19].tolist() == [0, 1, 1]                 # q-table indices of an i-frame (enc.rs:296-298)
    mv = rng.integers(-15, 16, (nb, 2)).astype(np.int8)
    mv[::3] = 0
    has = (rng.random(nb) < 0.6).astype(np.uint8)
    na = lib.pfv_serialize_pframe_payload(_p(mv), _p(has), _p(coef), nb, _p(a), cap)
    nbo = L.pfvo_serialize_pframe(_p(mv), _p(has), _p(coef), nb, _p(b), cap)
    assert na == nbo > 19 and np.array_equal(a[:na], b[:na])
    assert a[16:19].tolist() == [2, 3, 3]                 # enc.rs:409-411


def test_oversized_coefficient_is_rejected(graft, pkg):
    """|v| >= 16384 needs 16 size bits: the reference would index its 16-bin histogram out of range (rle.rs:44)"""
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    lib = pkg._lib.load()
    coef = np.zeros((1, 256), np.int16)
    coef[0, 0] = 16384
    assert lib.pfv_serialize_iframe_payload(_p(coef), 1, None, 0) == 0
    coef[0, 0] = 16383
    assert lib.pfv_serialize_iframe_payload(_p(coef), 1, None, 0) > 19


def test_oracle_stream_self_consistency(oracle, pkg):
    """oracle encoder -> .pfv bytes -> oracle decoder reproduces the encoder's reconstruction, incl. drop frames"""
    from oracle_bind import OracleStreamDecoder, OracleStreamEncoder
    w, h = 48, 32
    st = pkg.SyntheticStream(w, h)
    enc = OracleStreamEncoder(oracle, w, h, 25, 5)
    hot = oracle.encoder(w, h, 5)
    recon = []
    for t in range(5):
        f = st.frame(t)
        if t == 3:
            enc.encode_dropframe()
            continue
        if t == 0:
            enc.encode_iframe(f); hot.encode_iframe(f)
        else:
            enc.encode_pframe(f); hot.encode_pframe(f)
        pf = pkg.VideoFrame.from_packed(w, h, hot.prev_frame(), padded=True)
        recon.append(np.concatenate([pf.plane_y.image()[:h, :w].reshape(-1), pf.plane_u.image()[:h // 2, :w // 2].reshape(-1),
                                     pf.plane_v.image()[:h // 2, :w // 2].reshape(-1)]))
    enc.finish()
    data = enc.bytes()
    dec = OracleStreamDecoder(oracle, data)
    assert (dec.width, dec.height, dec.framerate) == (w, h, 25)
    out = []
    while True:
        rc, fr = dec.advance_frame()
        assert rc >= 0
        if fr is not None:
            out.append(fr)
        if rc == 0:
            break
    assert len(out) == 4 and all(np.array_equal(a, b) for a, b in zip(out, recon))
    for bad, code in ((b"X" + data[1:], -6), (data[:8] + (210).to_bytes(4, "little") + data[12:], -7), (data[:10], -8)):
        d = OracleStreamDecoder(oracle, bad)
        assert not d.h and d.err == code


@pytest.mark.parametrize("density", [0.0, 0.03, 0.2, 1.0])
def test_payload_parsers_invert_the_serialisers(graft, pkg, oracle, density):
    """product parse(serialise(x)) == x, dense and sparse forms, on payloads from the product and from the oracle;
    truncated payloads are reported, never read past"""
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    lib = pkg._lib.load()
    L = _oracle_payloads(oracle)
    rng = np.random.default_rng(int(density * 1000) + 1)
    nb = 41
    coef = (rng.integers(-16383, 16384, (nb, 256)) * (rng.random((nb, 256)) < density)).astype(np.int16)
    coef[7] = 0
    coef[9, 255] = 3
    mv = rng.integers(-15, 16, (nb, 2)).astype(np.int8)
    mv[::4] = 0
    has = (rng.random(nb) < 0.7).astype(np.uint8)
    cap = nb * 256 * 4 + 64
    for pframe in (False, True):
        a, b = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        if pframe:
            na = lib.pfv_serialize_pframe_payload(_p(mv), _p(has), _p(coef), nb, _p(a), cap)
            nbo = L.pfvo_serialize_pframe(_p(mv), _p(has), _p(coef), nb, _p(b), cap)
        else:
            na = lib.pfv_serialize_iframe_payload(_p(coef), nb, _p(a), cap)
            nbo = L.pfvo_serialize_iframe(_p(coef), nb, _p(b), cap)
        assert na == nbo
        want = coef * has[:, None].astype(np.int16) if pframe else coef
        for payload in (a[:na].copy(), b[:nbo].copy()):
            out, q = np.full((nb, 256), 77, np.int16), np.zeros(3, np.uint8)
            omv, ohas = np.zeros((nb, 2), np.int8), np.zeros(nb, np.uint8)
            rc = (lib.pfv_parse_pframe_payload(_p(payload), payload.size, nb, 4, _p(omv), _p(ohas), _p(out), _p(q)) if pframe else
                  lib.pfv_parse_iframe_payload(_p(payload), payload.size, nb, 4, _p(out), _p(q)))
            assert rc == 0 and np.array_equal(out, want) and q.tolist() == ([2, 3, 3] if pframe else [0, 1, 1])
            if pframe:
                assert np.array_equal(omv, mv) and np.array_equal(ohas, has)
            # sparse form: same non-zeros, ascending flat indices
            idx, val, n = np.zeros(nb * 256, np.uint32), np.zeros(nb * 256, np.int16), ctypes.c_size_t()
            rc = lib.pfv_parse_payload_sparse(int(pframe), _p(payload), payload.size, nb, 4, _p(omv), _p(ohas), _p(idx), _p(val),
                                              idx.size, ctypes.byref(n), _p(q))
            assert rc == 0
            dense = np.zeros(nb * 256, np.int16)
            dense[idx[:n.value]] = val[:n.value]
            assert np.array_equal(dense.reshape(nb, 256), want) and np.all(np.diff(idx[:n.value].astype(np.int64)) > 0)
            if n.value > 4:                                   # a list that is too short is reported, not overrun
                rc = lib.pfv_parse_payload_sparse(int(pframe), _p(payload), payload.size, nb, 4, _p(omv), _p(ohas), _p(idx), _p(val),
                                                  n.value - 1, ctypes.byref(n), _p(q))
                assert rc == 1
            # every truncation is an error (or, cut inside the final padding, still the full frame)
            for cut in (0, 10, 18, 19, payload.size // 2, payload.size - 1):
                out2 = np.zeros((nb, 256), np.int16)
                rc = (lib.pfv_parse_pframe_payload(_p(payload), cut, nb, 4, _p(omv), _p(ohas), _p(out2), _p(q)) if pframe else
                      lib.pfv_parse_iframe_payload(_p(payload), cut, nb, 4, _p(out2), _p(q)))
                assert rc in (pkg._lib.PFV_ERR_IO, pkg._lib.PFV_ERR_FORMAT) or (rc == 0 and np.array_equal(out2, want))
            # a q-table index past the header's table count fails at the head (dec.rs:244-246)
            bad = payload.copy(); bad[17] = 9
            rc = (lib.pfv_parse_pframe_payload(_p(bad), bad.size, nb, 4, _p(omv), _p(ohas), _p(out), _p(q)) if pframe else
                  lib.pfv_parse_iframe_payload(_p(bad), bad.size, nb, 4, _p(out), _p(q)))
            assert rc == pkg._lib.PFV_ERR_FORMAT


def test_stream_layout_regression_pins(pkg, oracle):
    """the oracle's .pfv bytes for four small clips still hash to the committed digests (tests/golden/stream_digests.json)"""
    import hashlib
    import json
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_stream_digests", os.path.join(sys_path, "make_stream_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pins = json.load(open(os.path.join(sys_path, "stream_digests.json")))
    assert [tuple(p["clip"]) for p in pins] == mod.CLIPS
    for p in pins:
        b = mod.stream_bytes(pkg, oracle, tuple(p["clip"]))
        assert len(b) == p["bytes"] and hashlib.sha256(b).hexdigest() == p["sha256"], p["clip"]
