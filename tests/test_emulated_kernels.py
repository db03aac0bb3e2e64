"""CPU-only logic check of the HIP kernel SOURCES: pretty-fast-video_amd/csrc/*.hip compiled
unmodified with g++ against tests/hipemu (every GPU thread a fiber, wavefront = 64) and driven
through the same C ABI and the same parity checks as the real-GPU tests, at tiny sizes.
This is not the parity gate (that is tests/test_gpu_parity.py on a real MI355X); it keeps
indexing / cross-lane / LDS-layout regressions from reaching the GPU box."""
import os

import numpy as np
import pytest

import parity_cases as pc
import stream_cases as sc

# Both lane mappings of the codec kernels (pfv_kernels.hip, "Lane mappings").  At these sizes the automatic choice is the
# small-grid mapping (16 lanes per macroblock); the kernel-level tests below run a second time with the batch mapping (8 lanes per
# macroblock) forced.  The stream-level tests (containers, batch objects) run once, on the automatic choice.
_BOTH_MAPPINGS = ("plane_ops", "golden", "trap", "session", "sparse_coded", "bad_motion", "gop_graph", "colour", "blit", "lists_decode")


# "lanes8split": the batch mapping with the p-frame encoder in its split form (k_pf_search + k_pf_transform, PFV_OPT_TILE_COMPACTION = 2), on the
# tests that encode p-frames
_SPLIT = ("plane_ops", "golden", "trap", "session", "sparse_coded", "gop_graph")


@pytest.fixture(autouse=True, params=["auto", "lanes8", "lanes8split"])
def lane_mapping(request, pkg, emu_ctx):
    L = pkg._lib
    if request.param != "auto":
        if not any(k in request.node.name for k in (_BOTH_MAPPINGS if request.param == "lanes8" else _SPLIT)):
            pytest.skip("stream-level test: automatic lane mapping only" if request.param == "lanes8" else "no p-frame encode in this test")
        emu_ctx.set_option(L.PFV_OPT_LANE_MAPPING, L.PFV_LANES_PER_MB_8)
        if request.param == "lanes8split":
            emu_ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 2)
    yield request.param
    emu_ctx.set_option(L.PFV_OPT_LANE_MAPPING, L.PFV_LANES_AUTO)
    emu_ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 1)


@pytest.mark.parametrize("w,h", [(64, 48), (50, 38), (144, 16), (400, 80)])   # 400x80 has interior strips (bounds-check-free search path)
@pytest.mark.parametrize("quality", [0, 5, 10])
def test_emu_plane_ops(pkg, emu_ctx, oracle, w, h, quality):
    il, ic, pl, pcq, px_err = oracle.qtables(quality)
    px = pc.smooth_plane(h, w, seed=w * 1000 + h + quality)
    pc.check_encode_plane(pkg, emu_ctx, oracle, px, il, 0)
    ref = pc.shifted_ref(px, 3, -2, seed=7, clear=128)
    pc.check_encode_plane_delta(pkg, emu_ctx, oracle, px, ref, pcq, px_err, 128)


def test_emu_sparse_coded_tiles(pkg, emu_ctx, oracle):
    """tile-level compaction of the coded macroblocks (skip-aware transform)"""
    assert pc.check_sparse_coded_tiles(pkg, emu_ctx, oracle, sizes=((256, 128), (130, 70))) == 48


def test_emu_golden_vectors(pkg, emu_ctx, oracle):
    pc.check_golden(pkg, emu_ctx, oracle)


def test_emu_trap_vectors(pkg, emu_ctx, oracle):
    pc.check_trap_vectors(pkg, emu_ctx, oracle)


def test_emu_session_two_streams(pkg, emu_ctx, oracle):
    stats = pc.check_session(pkg, emu_ctx, oracle, 64, 48, 5, n_streams=2, n_frames=3)
    assert 0 < stats["coded"] < stats["mbs"]


def test_emu_session_gop_batched(pkg, emu_ctx, oracle):
    """the GOPs of ONE stream in the slots of one session: 11 frames, GOP 4 -> slots of 4, 4 and 3 frames (window shrinks at step 3)"""
    r = pc.check_gop_batched_session(pkg, emu_ctx, oracle, 64, 48, 5, n_frames=11, gop=4)
    assert r == {"gops": 3, "launches_per_operation": 4, "frames": 11}
    pc.check_gop_batched_session(pkg, emu_ctx, oracle, 50, 38, 2, n_frames=6, gop=3)       # ragged planes: byte-wise crop path


def test_emu_gop_batched_clip(pkg, emu_ctx, oracle):
    """the whole-clip form of the same check (digests per frame, then the packets through pfv_gop_decoder with device entropy), toy size:
    23 frames, GOP 5 -> 5 slots, the last GOP short; decoded 2 GOPs per batch"""
    r = pc.check_gop_batched_clip(pkg, emu_ctx, oracle, 64, 48, 5, n_frames=23, gop=5, dec_gops=2)
    assert r["gops"] == 5 and r["frames"] == 23 and r["packets_read_on_device"] + r["packets_left_to_host_parser"] == 23


def test_emu_session_low_motion(pkg, emu_ctx, oracle):
    """static background + moving objects: tiles with few coded macroblocks take the compaction path, reconstruction included"""
    stats = pc.check_session(pkg, emu_ctx, oracle, 272, 144, 5, n_streams=2, n_frames=4, kind="low_motion")
    assert 0.05 < stats["coded"] / stats["mbs"] < 0.6


def test_emu_session_batched_dev(pkg, emu_ctx, oracle):
    """the benched call pattern (device-pointer entry points, device generator, fused crop) at toy size"""
    stats = pc.check_session_batched_dev(pkg, emu_ctx, oracle, 64, 48, 5, [pkg.synth.SEED + 17 * k for k in range(3)], n_frames=3)
    assert stats["mbs"] == 2 * 3 * 20 and 0 < stats["coded"]


def test_emu_bad_motion_vector(pkg, emu_ctx, oracle):
    q = oracle.qtables(5)[2]
    ref = pkg.VideoPlane(32, 32)
    src = pkg.EncodedPPlane(32, 32, 2, 2, np.array([[-1, 0], [0, 0], [0, 0], [0, 0]], np.int8), np.zeros(4, np.uint8),
                            np.zeros((4, 256), np.int16))
    with pytest.raises(pkg.PfvError) as e:
        pkg.VideoPlane.decode_plane_delta(src, ref, q, emu_ctx)
    assert e.value.code == pkg._lib.PFV_ERR_BAD_MV


def test_emu_stream_encoder_decoder(pkg, emu_ctx, oracle):
    """product Encoder/Decoder vs the oracle's, whole .pfv byte stream"""
    data = sc.check_stream_roundtrip(pkg, emu_ctx, oracle, 48, 32, 5, n_frames=5, gop=3, drop_at=(2,))
    sc.check_advance_delta(pkg, emu_ctx, oracle, data, kinds=[True, True, False, True, True])
    sc.check_header_errors(pkg, emu_ctx, data)


def test_emu_gop_graph(pkg, emu_ctx, oracle):
    pc.check_gop_graph(pkg, emu_ctx, oracle, n_frames=3)


def test_emu_empty_pframe_packet_is_an_error(pkg, emu_ctx, oracle):
    sc.check_empty_pframe_packet(pkg, emu_ctx, oracle)


def test_emu_encoder_keeps_nothing(pkg, emu_ctx):
    sc.check_encoder_keeps_nothing(pkg, emu_ctx)


def test_emu_colour_utils(pkg, emu_ctx, oracle):
    pc.check_colour_utils(pkg, emu_ctx, oracle)


def test_emu_blit_dev(pkg, emu_ctx, oracle):
    assert pc.check_blit_dev(pkg, emu_ctx, oracle, n_random=12) >= 50


def test_emu_misaligned_device_frames(pkg, emu_ctx, oracle):
    pc.check_misaligned_device_frames(pkg, emu_ctx, oracle)


def test_emu_fuzz_plane_operators(pkg, emu_ctx, oracle):
    pc.fuzz_plane_ops(pkg, emu_ctx, oracle, n_cases=12, seed=5, max_w=200, max_h=90)


def test_emu_corrupted_streams(pkg, emu_ctx, oracle):
    """hostile .pfv bytes: product parser + kernels vs the oracle's, outcome by outcome"""
    data, _ = sc.encode_clip(pkg, emu_ctx, oracle, 48, 32, 30, 5, n_frames=4, gop=2)
    stats = sc.check_corrupted_streams(pkg, emu_ctx, oracle, data, n_trials=160, seed=5)
    assert stats["trials"] == 160
    sc.check_lookahead_reset(pkg, emu_ctx, data, n_frames=4)


def test_emu_device_entropy(pkg, emu_ctx, oracle):
    """k_ent_* (RLE + Huffman + bit packing on the device) vs the oracle's packet serialisers, byte for byte"""
    assert pc.check_device_entropy(pkg, emu_ctx, oracle, 48, 32, n_streams=2, seed=3) == 20
    assert pc.check_device_entropy(pkg, emu_ctx, oracle, 34, 18, n_streams=1, seed=4, kinds=("typical", "edges")) == 4


def test_emu_lists_decode(pkg, emu_ctx):
    """coefficient lists expanded in the decode kernels' LDS stage == the dense arrays (both lane mappings); 144 x 16: a strip of 9 macroblocks
    and chroma strips of 5 (ragged wavefronts); 48 x 32: whole strips"""
    assert pc.check_lists_decode(pkg, emu_ctx, 48, 32, n_streams=2) > 0
    assert pc.check_lists_decode(pkg, emu_ctx, 144, 16, n_streams=1, seed=14, kinds=("dense", "typical", "edges")) > 0


def test_emu_sparse_decode(pkg, emu_ctx):
    pc.check_sparse_decode(pkg, emu_ctx, 48, 32, n_streams=2)


def test_emu_dense_stream_falls_back(pkg, emu_ctx, oracle):
    """white noise at quality 10: more than 1 coefficient in 4 is non-zero, the decoder's sparse list overflows and the
    packet is re-parsed into the dense form"""
    import io
    from oracle_bind import OracleStreamDecoder
    w, h = 48, 32
    rng = np.random.default_rng(2)
    buf = io.BytesIO()
    enc = pkg.Encoder(buf, w, h, 30, 10, emu_ctx)
    for t in range(3):
        fr = sc.frame_of(pkg, w, h, rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8))
        (enc.encode_iframe if t == 0 else enc.encode_pframe)(fr)
    enc.finish(); enc.close()
    data = buf.getvalue()
    assert len(data) > w * h * 3 // 2          # dense indeed: bigger than a raw frame
    a = [x for x in sc._outcomes_product(pkg, emu_ctx, data, lookahead=1)]
    b = [x for x in sc._outcomes_oracle(oracle, data)]
    assert a == b and sum(1 for x in a if x[0] == "frame") == 3


def test_emu_async_entropy_api(pkg, emu_ctx, oracle):
    """call sequence of the two-stream entropy mode (the emulator has one timeline; ordering is checked on the GPU)"""
    pc.check_async_entropy(pkg, emu_ctx, oracle, 48, 32, n_streams=2, n_frames=5)


def test_emu_colour_conversions(pkg, emu_ctx, oracle):
    pc.check_colour_conversions(pkg, emu_ctx, oracle, exhaustive=False)


def test_emu_batch_encoder(pkg, emu_ctx, oracle):
    sc.check_batch_encoder(pkg, emu_ctx, oracle, 48, 32, 5, n_streams=3, n_frames=4, gop=3)


def test_emu_batch_encoder_writer_failure_is_reported(pkg, emu_ctx):
    sc.check_batch_encoder_writer_failure(pkg, emu_ctx)


def test_emu_batch_encoder_dense_content_grows_its_buffer(pkg, emu_ctx, oracle):
    """white noise at quality 10 needs more payload bytes than a raw frame: the first download reports NOMEM, the
    encoder retries with the worst-case buffer and still writes the oracle's bytes"""
    import io
    from oracle_bind import OracleStreamEncoder
    w, h, n = 64, 48, 2
    rng = np.random.default_rng(8)
    bufs = [io.BytesIO() for _ in range(n)]
    enc = pkg.BatchEncoder(bufs, w, h, 30, 10, emu_ctx)
    enc._payloads = emu_ctx.host_array(4096)                      # start far too small
    oencs = [OracleStreamEncoder(oracle, w, h, 30, 10) for _ in range(n)]
    for t in range(2):
        for s_ in range(n):
            f = rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8)
            enc.frames[s_] = f
            (oencs[s_].encode_iframe if t == 0 else oencs[s_].encode_pframe)(f)
        (enc.encode_iframes if t == 0 else enc.encode_pframes)()
    enc.finish(); enc.close()
    for s_ in range(n):
        oencs[s_].finish()
        assert bufs[s_].getvalue() == oencs[s_].bytes()


def test_emu_batch_decoder(pkg, emu_ctx, oracle):
    sc.check_batch_decoder(pkg, emu_ctx, oracle, 48, 32, 5, n_streams=3, n_frames=4, gop=3)


def test_emu_batch_decoder_dense_fallback(pkg, emu_ctx, oracle):
    """quality 10 on small frames: more than 1 coefficient in 4 is non-zero, the step is parsed into the dense form"""
    sc.check_batch_decoder(pkg, emu_ctx, oracle, 48, 32, 10, n_streams=2, n_frames=3, gop=2, noise=True)


def test_emu_gop_objects(pkg, emu_ctx, oracle):
    """pfv_gop_encoder / pfv_gop_decoder: same bytes / frames as the frame-by-frame objects and the oracle, whatever the batch shape"""
    # three GOPs of 4 + drop frames; shapes: everything in one batch / batches of 2 groups / groups cut after 3 frames (runs continue
    # across batches) / one slot (degenerates to the serial order)
    data = sc.check_gop_objects(pkg, emu_ctx, oracle, 64, 48, 5, "IPPPIPDPPIPPP", shapes=((8, 15), (2, 3)), alternate_modes=True)
    assert data[-5:] == bytes(5)
    # a stream that starts with p-frames (prev_frame = new_padded, src/enc.rs:46) and has GOPs of unequal length
    sc.check_gop_objects(pkg, emu_ctx, oracle, 50, 38, 3, "PPIPIPDIP", shapes=((1, 2),), alternate_modes=True)


def test_emu_gop_decoder_device_entropy(pkg, emu_ctx, oracle):
    """k_entd_*: payloads read by the self-synchronising device stage; unsettled / periodic / long-code content"""
    only = None
    out = sc.check_gop_device_entropy(pkg, emu_ctx, oracle, 96, 64, pattern="IPPIP", only=only)
    assert out["noise"]["packets_read_on_device"] >= 1, out


def test_emu_device_block_headers(pkg, emu_ctx, oracle):
    """k_hdr_*: 272 x 144 (153 + 2 x 45 macroblocks: up to 3 888 header bits = 2 header workgroups), mixed 2- and 16-bit headers"""
    assert sc.check_device_block_headers(pkg, emu_ctx, oracle, 272, 144, pattern="IPP") == {"low_motion": 3, "pan": 3}


def test_emu_one_symbol_table_lists(pkg, emu_ctx, oracle):
    """a degenerate code table whose values outnumber what the list pool set aside for the packet's bits: host parser, second parse, spill buffer"""
    assert sc.check_one_symbol_table_lists(pkg, emu_ctx, oracle) == 5


def test_emu_gop_decoder_device_entropy_small_stages():
    """ONE variant build with every staging buffer shrunk, re-running the checks that lean on them:
    k_ent_pack's window holds 64 words instead of 2048 (test_emu_device_entropy: the dense and typical cases re-anchor the window several
    times per group and single steps overflow it -- the memory path); k_entd_emit stages 8 entries and 2 macroblock starts per workgroup
    instead of 4096 / 1024 (everything behind them goes to memory directly: the paths content far denser than any real frame takes); the
    header scan takes one workgroup map per tile"""
    import subprocess
    import sys
    env = dict(os.environ, PFV_EMU_DEFS="-DPFV_ENT_WIN_WORDS=64 -DPFV_ED_OUT_CAP=8 -DPFV_ED_MB_CAP=2 -DPFV_HDR_SCAN_TILE=1", PFV_TEST_VARIANT_BUILD="1")
    me = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        me + "::test_emu_device_entropy", me + "::test_emu_gop_decoder_device_entropy", me + "::test_emu_device_block_headers"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout


def test_emu_gop_encoder_flush_and_errors(pkg, emu_ctx, oracle):
    sc.check_gop_encoder_flush_and_errors(pkg, emu_ctx, oracle)


def test_emu_gop_decoder_corrupted_streams(pkg, emu_ctx, oracle):
    data, _ = sc.encode_pattern(pkg, emu_ctx, oracle, 48, 32, 5, "IPPIPPPIP", lambda buf: pkg.Encoder(buf, 48, 32, 30, 5, emu_ctx), with_oracle=False)
    stats = sc.check_gop_decoder_corrupted(pkg, emu_ctx, oracle, data, n_trials=36, seed=4)
    assert stats["trials"] == 36 and stats["errors"] > 1 and stats["frames_after_an_error"] > 0


def test_emu_gop_decoder_dense_iframe_failure(pkg, emu_ctx, oracle):
    assert sc.check_gop_decoder_dense_iframe_failure(pkg, emu_ctx, oracle, shapes=((8, 15), (2, 2), (1, 15))) >= 1


def test_emu_soak_iterations(pkg, emu_ctx, oracle, lane_mapping):
    """a few passes of tools/soak.py's randomised iteration (every checker, random geometry / quality / packet pattern / lane mapping /
    option settings) at sizes the emulator finishes in seconds; fixed seeds.  The GPU box runs the same function at full sizes for minutes."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak
    L = pkg._lib
    stats = {"plane_cases": 0, "sessions": 0, "entropy_payloads": 0, "corrupted_trials": 0, "stream_roundtrips": 0, "batch": 0}
    try:
        for it, seed in enumerate((11, 12, 13, 14)):
            soak.iteration(pkg, emu_ctx, oracle, seed, it, stats, small=True)
    finally:
        emu_ctx.set_option(L.PFV_OPT_LANE_MAPPING, L.PFV_LANES_AUTO)
        emu_ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 1)
        emu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_AUTO)
    assert stats["sessions"] == 4 and stats["gop_batched"] == 4 and stats["corrupted_trials"] == 48
