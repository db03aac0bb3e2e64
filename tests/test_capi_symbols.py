"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol that include/pfv_hip.h
declares; without a GPU the product fails loudly instead of falling back to anything."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("pfv_hip_core.h", "pfv_hip_ext.h")       # pfv_hip.h includes both and declares nothing itself


def declared_symbols(headers=HEADERS):
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in headers)
    return sorted(set(re.findall(r"PFV_API\s+[\w\s\*]+?\b(pfv_\w+)\s*\(", text)))


def test_header_tiers():
    """pfv_hip_core.h = the drop-in boundary (what INTEGRATION.md binds): the six plane operators, q-tables, sessions, the frame-at-a-time
    stream objects, context, geometry -- host pointers only, nothing else; pfv_hip.h declares nothing of its own; both tiers are valid C99"""
    import subprocess
    core = declared_symbols(("pfv_hip_core.h",))
    assert {"pfv_encode_plane", "pfv_encode_plane_delta", "pfv_decode_plane_into", "pfv_decode_plane_delta", "pfv_decode_plane_delta_into",
            "pfv_qtables_from_quality", "pfv_enc_session_create", "pfv_enc_iframe", "pfv_enc_pframe", "pfv_dec_session_create", "pfv_dec_iframe",
            "pfv_dec_pframe", "pfv_dec_get_frame", "pfv_encoder_create", "pfv_encoder_encode_pframe", "pfv_decoder_create",
            "pfv_decoder_advance_frame", "pfv_ctx_create"} <= set(core)
    assert not [n for n in core if n.endswith("_dev") or n.startswith(("pfv_gop_", "pfv_batch_", "pfv_comm_", "pfv_graph_", "pfv_event_", "pfv_synth_"))], core
    assert len(core) <= 48
    assert not set(core) & set(declared_symbols(("pfv_hip_ext.h",)))
    assert not declared_symbols(("pfv_hip.h",))
    for h in ("pfv_hip_core.h", "pfv_hip.h"):
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", os.path.join(ROOT, "include", h)], check=True)
    # every symbol INTEGRATION.md binds from Rust is a core symbol
    bound = set(re.findall(r"\bfn (pfv_\w+)\(", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert bound and bound <= set(core), bound - set(core)


def test_header_symbols_are_all_bound_and_exported(graft, pkg):
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    lib = pkg._lib.load()                       # binds every entry of SIGNATURES; AttributeError if one is missing
    assert pkg._lib._lib_path == pkg._lib.DEFAULT_LIB
    declared = declared_symbols()
    bound = sorted(name for name, _, _ in pkg._lib.SIGNATURES)
    assert declared == bound, f"header vs binding mismatch: {set(declared) ^ set(bound)}"
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.pfv_version()
    # the binary is attributable to the sources in the tree: build_hip() stamps their hash into pfv_version()
    assert graft.source_hash().encode() in lib.pfv_version() and graft.hip_build_id() == graft.source_hash()
    assert lib.pfv_pad16(1080) == 1088 and lib.pfv_pad16(1920) == 1920
    assert lib.pfv_total_blocks(1920, 1080) == 12240 and lib.pfv_total_blocks(3840, 2160) == 48720
    assert lib.pfv_frame_bytes(1920, 1080) == 3110400 and lib.pfv_padded_frame_bytes(1920, 1080) == 3133440


def test_oracle_is_not_reachable_from_the_product():
    """nothing under the package or the C sources refers to oracle/"""
    pkgdir = os.path.join(ROOT, "pretty-fast-video_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pfv_oracle" not in text and "oracle_bind" not in text and "pfvo_" not in text, f


def test_no_gpu_fails_loudly(graft, pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    with pytest.raises(pkg.PfvError) as e:
        pkg.Context(0)
    assert e.value.code in (pkg._lib.PFV_ERR_NO_DEVICE, pkg._lib.PFV_ERR_HIP)


def test_qtables_entry_point_matches_oracle(graft, pkg, oracle):
    graft.build_hip()
    __import__("libswitch").reset(pkg)
    import numpy as np
    for quality in range(11):
        got = pkg.qtables_from_quality(quality)
        want = oracle.qtables(quality)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b)
        assert got[4] == want[4]
    with pytest.raises(pkg.PfvError):
        pkg.qtables_from_quality(11)
