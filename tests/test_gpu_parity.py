"""Parity tests proper: the HIP kernels on a real MI355X, called through the C ABI
(libpfv_hip.so), against the CPU oracle on the same seeded inputs.  Bit-exact everywhere
(integer / byte / index work)."""
import ctypes
import os

import numpy as np
import pytest

import parity_cases as pc
import stream_cases as sc

pytestmark = pytest.mark.gpu


# tests that encode p-frames: they run a third time with the p-frame encoder in its split form (k_pf_search + k_pf_transform)
_PFRAME_TESTS = ("golden", "gop_graph", "trap", "pframe", "session", "benched_shape", "gop_batched", "config5", "sparse_coded", "fuzz", "extreme_aspect", "misaligned")


@pytest.fixture(autouse=True, params=["lanes8", "lanes16", "lanes8split"])
def lane_mapping(request, pkg, gpu_ctx):
    """every test of this file runs under BOTH lane mappings of the codec kernels (pfv_kernels.hip, "Lane mappings": 8 lanes per
    macroblock = the batch mapping, 16 = the small-grid mapping the library picks for launches of fewer than 4 096 strips), forced
    through pfv_ctx_set_option; sessions pick the option up when they are created.  lanes8split: the batch mapping with the p-frame
    encoder as two kernels (PFV_OPT_TILE_COMPACTION = 2), on the tests that encode p-frames."""
    L = pkg._lib
    if request.param == "lanes8split":
        if not any(k in request.node.name for k in _PFRAME_TESTS):
            pytest.skip("no p-frame encode in this test")
        gpu_ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 2)
    gpu_ctx.set_option(L.PFV_OPT_LANE_MAPPING, L.PFV_LANES_PER_MB_16 if request.param == "lanes16" else L.PFV_LANES_PER_MB_8)
    yield request.param
    gpu_ctx.set_option(L.PFV_OPT_LANE_MAPPING, L.PFV_LANES_AUTO)
    gpu_ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 1)


def test_native_library_is_the_one_loaded(pkg, gpu_ctx):
    """the in-tree gfx950 build must be what runs (no emulator, no fallback)"""
    assert pkg._lib._lib_path == pkg._lib.DEFAULT_LIB
    with open("/proc/self/maps") as f:
        assert "libpfv_hip.so" in f.read()
    assert b"gfx950" in pkg._lib.load().pfv_version()
    import __graft_entry__ as graft
    assert graft.source_hash().encode() in pkg._lib.load().pfv_version(), "libpfv_hip.so was not built from the sources in this tree"


def test_golden_vectors(pkg, gpu_ctx, oracle):
    pc.check_golden(pkg, gpu_ctx, oracle)


def test_gop_graph_replay(pkg, gpu_ctx, oracle):
    pc.check_gop_graph(pkg, gpu_ctx, oracle, 64, 48, n_streams=2, n_frames=4)
    pc.check_gop_graph(pkg, gpu_ctx, oracle, 320, 240, n_streams=1, n_frames=15)


def test_trap_vectors(pkg, gpu_ctx, oracle):
    pc.check_trap_vectors(pkg, gpu_ctx, oracle)


@pytest.mark.parametrize("quality", list(range(0, 11)))
def test_iframe_plane_all_qualities(pkg, gpu_ctx, oracle, quality):
    il, ic, _, _, _ = oracle.qtables(quality)
    px = pc.smooth_plane(96, 208, seed=quality)
    pc.check_encode_plane(pkg, gpu_ctx, oracle, px, il, 0)
    pc.check_encode_plane(pkg, gpu_ctx, oracle, px, ic, 128)


@pytest.mark.parametrize("w,h", [(16, 16), (50, 38), (128, 16), (130, 18), (144, 160), (1, 1), (17, 33), (960, 540)])
def test_iframe_plane_ragged_sizes(pkg, gpu_ctx, oracle, w, h):
    il = oracle.qtables(5)[0]
    rng = np.random.default_rng(w * 7 + h)
    px = rng.integers(0, 256, (h, w), dtype=np.uint8)          # uniform noise: worst case for the quantiser
    pc.check_encode_plane(pkg, gpu_ctx, oracle, px, il, 77)


@pytest.mark.parametrize("fill", ["zeros", "ones", "checker", "checker8", "ramp"])
def test_iframe_extreme_blocks(pkg, gpu_ctx, oracle, fill):
    h, w = 64, 160
    y, x = np.mgrid[0:h, 0:w]
    px = {"zeros": np.zeros((h, w)), "ones": np.full((h, w), 255), "checker": ((x + y) & 1) * 255,
          "checker8": (((x >> 3) + (y >> 3)) & 1) * 255, "ramp": (x * 3 + y * 5) & 255}[fill].astype(np.uint8)
    for quality in (0, 1, 5, 10):
        il = oracle.qtables(quality)[0]
        pc.check_encode_plane(pkg, gpu_ctx, oracle, px, il, 0)


def test_iframe_arbitrary_qtable_and_coefficients(pkg, gpu_ctx, oracle):
    """decode side with hostile input: random i16 coefficients (i32 wrap-around in the iDCT)
    and q entries up to 65535"""
    rng = np.random.default_rng(3)
    q = rng.integers(1, 65536, 64).astype(np.int32)
    bw, bh = 9, 3
    coef = rng.integers(-32768, 32768, (bw * bh, 256)).astype(np.int16)
    src = pkg.EncodedIPlane(bw * 16, bh * 16, bw, bh, coef)
    dec = pkg.VideoPlane.decode_plane(src, q, gpu_ctx)
    assert np.array_equal(dec.image(), oracle.decode_plane(coef, bw, bh, q))
    # encode side with large q
    px = rng.integers(0, 256, (48, 144), dtype=np.uint8)
    q2 = rng.integers(1, 300, 64).astype(np.int32)
    pc.check_encode_plane(pkg, gpu_ctx, oracle, px, q2, 0)


@pytest.mark.parametrize("quality", [0, 1, 2, 5, 8, 10])
@pytest.mark.parametrize("w,h,dx,dy", [(208, 96, 5, -3), (50, 38, -7, 2), (272, 48, 15, 15), (144, 160, -15, -9), (528, 112, 9, -14)])
def test_pframe_plane(pkg, gpu_ctx, oracle, quality, w, h, dx, dy):
    _, _, pl, pcq, px_err = oracle.qtables(quality)
    px = pc.smooth_plane(h, w, seed=quality * 31 + w)
    ref = pc.shifted_ref(px, dx, dy, seed=quality + h, clear=128)
    enc, _ = pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, px, ref, pl, px_err, 128)
    pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, px, ref, pcq, px_err, 0)
    if quality >= 5:
        assert enc.has_coeff.sum() < enc.has_coeff.size      # the skip path ran too


def test_pframe_sparse_coded_tiles(pkg, gpu_ctx, oracle):
    """tile-level compaction of the coded macroblocks (skip-aware transform), both encoder forms"""
    assert pc.check_sparse_coded_tiles(pkg, gpu_ctx, oracle) == 96
    L = pkg._lib
    gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_INT)
    try:
        assert pc.check_sparse_coded_tiles(pkg, gpu_ctx, oracle, sizes=((256, 128), (130, 70))) == 48
    finally:
        gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_AUTO)


def test_pframe_noise_and_ties(pkg, gpu_ctx, oracle):
    """white noise (no gradient: many near-ties), a flat plane (exact ties everywhere: the
    centre must win) and a reference identical to the source (zero error)"""
    _, _, pl, _, px_err = oracle.qtables(5)
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (80, 176), dtype=np.uint8)
    pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, noise, rng.integers(0, 256, (80, 176), dtype=np.uint8), pl, px_err, 0)
    flat = np.full((80, 176), 90, np.uint8)
    enc, _ = pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, flat, np.full((80, 176), 93, np.uint8), pl, 0.0, 0)
    assert not enc.motion.any()
    enc, _ = pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, noise, noise.copy(), pl, px_err, 0)
    assert not enc.motion.any() and not enc.has_coeff.any()
    # periodic pattern: many exactly equal candidates, first-visited must win
    y, x = np.mgrid[0:80, 0:176]
    per = (((x >> 2) + (y >> 2)) & 1).astype(np.uint8) * 200
    pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, per, np.roll(per, (4, 4), (0, 1)), pl, 0.0, 0)


def test_pframe_residual_extremes(pkg, gpu_ctx, oracle):
    """+-255 residuals (clamp and halving paths)"""
    _, _, pl, _, _ = oracle.qtables(1)
    y, x = np.mgrid[0:48, 0:144]
    a = ((((x >> 1) + (y >> 1)) & 1) * 255).astype(np.uint8)
    pc.check_encode_plane_delta(pkg, gpu_ctx, oracle, a, (255 - a).astype(np.uint8), pl, 0.0, 0)


def test_bad_motion_vector_is_an_error(pkg, gpu_ctx, oracle):
    q = oracle.qtables(5)[2]
    ref = pkg.VideoPlane(32, 32)
    for mvbad in ([-1, 0], [0, -1], [17, 0], [0, 17]):
        mv = np.zeros((4, 2), np.int8)
        mv[0 if mvbad[0] < 0 or mvbad[1] < 0 else 3] = mvbad
        src = pkg.EncodedPPlane(32, 32, 2, 2, mv, np.zeros(4, np.uint8), np.zeros((4, 256), np.int16))
        with pytest.raises(pkg.PfvError) as e:
            pkg.VideoPlane.decode_plane_delta(src, ref, q, gpu_ctx)
        assert e.value.code == pkg._lib.PFV_ERR_BAD_MV


def test_bad_arguments(pkg, gpu_ctx, oracle):
    plane = pkg.VideoPlane(32, 32)
    q = np.zeros(64, np.int32)                        # q == 0: the reference divides by zero (panic)
    with pytest.raises(pkg.PfvError) as e:
        plane.encode_plane(q, 0, gpu_ctx)
    assert e.value.code == pkg._lib.PFV_ERR_BAD_ARG
    with pytest.raises(pkg.PfvError):
        pkg.EncoderSession(gpu_ctx, 63, 48, 5, 1)     # odd width (src/frame.rs:13)
    with pytest.raises(pkg.PfvError):
        pkg.EncoderSession(gpu_ctx, 64, 48, 11, 1)    # quality > 10 (src/enc.rs:38)


def test_session_small_gop_multistream(pkg, gpu_ctx, oracle):
    stats = pc.check_session(pkg, gpu_ctx, oracle, 176, 144, 5, n_streams=3, n_frames=6, gop=4)
    assert 0 < stats["coded"] < stats["mbs"]


def test_session_low_motion_content(pkg, gpu_ctx, oracle):
    """static background + moving objects (~25 % coded): tiles with a few coded macroblocks take the skip-aware (compacted)
    transform; closed-loop reconstruction, decoder and cropped output against the oracle -- small, 1080p, and the batched
    device-pointer form; both encoder forms"""
    stats = pc.check_session(pkg, gpu_ctx, oracle, 640, 360, 5, n_streams=3, n_frames=6, gop=4, kind="low_motion")
    assert 0.1 < stats["coded"] / stats["mbs"] < 0.5
    stats = pc.check_session(pkg, gpu_ctx, oracle, 1920, 1080, 5, n_streams=1, n_frames=3, threads=os.cpu_count() or 1, kind="low_motion")
    assert 0.15 < stats["coded"] / stats["mbs"] < 0.45
    for q in (2, 8):
        pc.check_session(pkg, gpu_ctx, oracle, 400, 200, q, n_streams=2, n_frames=3, kind="low_motion")
    seeds = [pkg.synth.SEED + 17 * k for k in range(8)]
    pc.check_session_batched_dev(pkg, gpu_ctx, oracle, 1920, 1080, 5, seeds, n_frames=2, threads=min(32, len(os.sched_getaffinity(0))), kind="low_motion")
    oracle.L.pfvo_pool_shutdown()
    L = pkg._lib
    gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_INT)
    try:
        pc.check_session(pkg, gpu_ctx, oracle, 640, 360, 5, n_streams=2, n_frames=3, kind="low_motion")
    finally:
        gpu_ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_AUTO)


def test_session_odd_chroma_geometry(pkg, gpu_ctx, oracle):
    """width/2 not a multiple of 16 (chroma padded independently, src/frame.rs:31-36) and
    sizes whose chroma rows are not 16-byte aligned (slow-path source loads)"""
    pc.check_session(pkg, gpu_ctx, oracle, 100, 60, 2, n_streams=2, n_frames=3, gop=15)
    pc.check_session(pkg, gpu_ctx, oracle, 18, 2, 10, n_streams=1, n_frames=2, gop=15)


def test_session_1080p_iframe_and_pframe(pkg, gpu_ctx, oracle):
    """BASELINE configs #2 and #3 at full size: one 1080p i-frame round trip and one p-frame
    encode, every byte compared with the oracle (6 266 880 B of coefficients per frame)."""
    stats = pc.check_session(pkg, gpu_ctx, oracle, 1920, 1080, 5, n_streams=1, n_frames=2, threads=os.cpu_count() or 1)
    assert stats["mbs"] == 12240
    assert 0 < stats["coded"] < stats["mbs"]


def test_benched_shape_96_streams_1080p_vs_oracle(pkg, gpu_ctx, oracle):
    """The shape bench.py times -- 96 distinct 1080p streams (its own seed table) per launch, device-pointer entry points,
    fused retframe crop, quality 5 -- one i-frame and one p-frame step, every byte of all 96 streams against the oracle
    (2 x 1 175 040 macroblocks; 601 MB of coefficients per step)."""
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    table = shard.assign_streams(n_streams_total=96, world=1, base_seed=pkg.synth.SEED)
    seeds = [int(r[1]) for r in shard.streams_of_rank(table, 0)]
    assert len(seeds) == 96 and len(set(seeds)) == 96
    threads = min(32, len(os.sched_getaffinity(0)))
    stats = pc.check_session_batched_dev(pkg, gpu_ctx, oracle, 1920, 1080, 5, seeds, n_frames=2, threads=threads)
    oracle.L.pfvo_pool_shutdown()
    assert stats["mbs"] == 96 * 12240 and 0 < stats["coded"] < stats["mbs"]


@pytest.mark.parametrize("geom", [(3840, 2160, 31), (1920, 1080, 61)])
def test_gop_batched_session_vs_serial_oracle(pkg, gpu_ctx, oracle, geom):
    """GOP-batched sessions (BASELINE configs #4 / #5 as bench.py --workload config5 runs them): the GOPs of ONE stream in the slots of a
    launch -- 4K x 31 frames (GOPs of 15, 15 and 1: the window shrinks after the i-frame step) and 1080p x 61 frames (five GOPs) -- with
    every coefficient, header, device-built packet payload and display-order decoded frame against the oracle's SERIAL encoder."""
    w, h, n = geom
    threads = min(32, len(os.sched_getaffinity(0)))
    r = pc.check_gop_batched_session(pkg, gpu_ctx, oracle, w, h, 5, n_frames=n, gop=15, threads=threads)
    oracle.L.pfvo_pool_shutdown()
    assert r["gops"] == (n + 14) // 15 and r["launches_per_operation"] == 15 and r["frames"] == n


def test_config5_launch_shape_4k_300_frames_vs_oracle(pkg, gpu_ctx, oracle):
    """The launch shape `bench.py --workload config5` times, at its size: one 3840x2160 stream of 300 frames, GOP 15, all 20 GOPs in the slots
    of a launch (20 x 48 720 macroblocks per frame operation) -- every frame's coefficients, motion vectors, skip flags, device-built packet
    payload and display-order decoded frame against the oracle's SERIAL encoder (BLAKE2 digests per frame), then the 300 packets as a .pfv
    stream through pfv_gop_decoder, 20 GOPs per batch, payloads read by the device's entropy stage: every delivered frame against the
    oracle's reconstruction (BASELINE configs #4 / #5; src/enc.rs:84-97, README.md:34-41)."""
    threads = min(32, len(os.sched_getaffinity(0)))
    r = pc.check_gop_batched_clip(pkg, gpu_ctx, oracle, 3840, 2160, 5, n_frames=300, gop=15, threads=threads)
    oracle.L.pfvo_pool_shutdown()
    assert r["gops"] == 20 and r["frames"] == 300
    assert r["packets_read_on_device"] == 300, r        # the synthetic content settles: nothing is left to the host parser


def test_gop_batched_session_low_motion_and_static(pkg, gpu_ctx, oracle):
    """the same with content that takes the skip-aware paths inside the batch (tile compaction, wavefronts with nothing coded)"""
    pc.check_gop_batched_session(pkg, gpu_ctx, oracle, 640, 368, 5, n_frames=22, gop=5, kind="low_motion", threads=8)
    pc.check_gop_batched_session(pkg, gpu_ctx, oracle, 336, 256, 8, n_frames=9, gop=4, kind="static", threads=8)


def test_rccl_world_of_every_visible_gpu(pkg, gpu_ctx):
    """the real world > 1 RCCL path (ncclCommInitRank over xGMI, the table broadcast, barriers and the counter reduction of
    bench.py) wherever at least two GPUs are visible -- one rank per visible device, self-launched like `python bench.py --gpus N`.
    The build pool's boxes have ONE MI355X: there this test is skipped, and says so."""
    import json
    import subprocess
    import sys
    n = int(gpu_ctx._lib.pfv_device_count())
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-rank RCCL path needs at least two (the 1-rank communicator test below runs here)")
    n = min(n, 8)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--streams", "8", "--no-entropy"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    res = json.loads(line[0])
    assert res["n_gpus"] == n and res["rccl_ranks"] == n and res["control_plane"]["backend"] == "rccl", res["control_plane"]
    ranks = res["control_plane"]["ranks"]
    assert sorted(x["device_ordinal"] for x in ranks) == list(range(n)) and len({x["pci_bus_id"] for x in ranks}) == n
    assert abs(sum(x["macroblocks_per_s"] * x["seconds"] for x in ranks) - n * 8 * 15 * 12240 * 2) < 1.0


def test_rccl_one_rank_communicator(pkg, gpu_ctx):
    """the control plane of the multi-GPU path on the real RCCL: a 1-rank communicator on the MI355X box runs the very entry
    points the 8-GPU job uses (pfv_comm_*: ncclCommInitRank, ncclBroadcast, ncclAllReduce, ncclAllGather on the context's
    stream, device buffers) -- through comm.Comm as bench.py drives it, and directly"""
    from importlib import import_module
    commlib = import_module("pretty_fast_video_amd.comm")
    shard = import_module("pretty_fast_video_amd.shard")
    rdzv = commlib.Rendezvous(0, 1)
    comm = commlib.Comm(gpu_ctx, rdzv, use_rccl=True)
    assert comm.backend == "rccl" and comm.handle is not None
    lib = gpu_ctx._lib
    assert lib.pfv_comm_rank(comm.handle) == 0 and lib.pfv_comm_world(comm.handle) == 1
    table = shard.assign_streams(96, 1, pkg.synth.SEED)
    assert np.array_equal(comm.broadcast_array(table), table)
    assert comm.allreduce([1175040.0 * 15, 3.5], "sum").tolist() == [1175040.0 * 15, 3.5]
    assert comm.allreduce([0.0123, -7.0], "max").tolist() == [0.0123, -7.0]
    comm.barrier()
    # device-buffer forms
    src = np.arange(256, dtype=np.uint8)
    d_a, d_b = gpu_ctx.alloc(256), gpu_ctx.alloc(256)
    gpu_ctx.upload(d_a, src)
    gpu_ctx.check(lib.pfv_comm_allgather_dev(comm.handle, ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), 256))
    out = np.empty(256, np.uint8)
    gpu_ctx.download(out, d_b)
    assert np.array_equal(out, src)
    vals = np.array([1.5, 2.5, -3.0])
    gpu_ctx.upload(d_a, vals)
    gpu_ctx.check(lib.pfv_comm_allreduce_f64_dev(comm.handle, ctypes.c_void_p(d_a), 3, pkg._lib.PFV_COMM_MAX))
    back = np.empty(3)
    gpu_ctx.download(back, d_a)
    assert back.tolist() == [1.5, 2.5, -3.0]
    assert lib.pfv_comm_broadcast_dev(comm.handle, ctypes.c_void_p(d_a), 24, 5) == pkg._lib.PFV_ERR_BAD_ARG      # root outside the job
    gpu_ctx.free(d_a); gpu_ctx.free(d_b)
    comm.close()
    rdzv.close()


def test_context_stream_priorities(pkg, oracle):
    """pfv_ctx_create_prio: contexts on the device's greatest / least stream priority produce the same bytes (an encoder on the
    high one, a decoder on the low one, ordered by pfv_ctx_wait_event: the two-context schedule of bench.py)"""
    with pkg.Context(0, priority=1) as ectx, pkg.Context(0, priority=-1) as dctx:
        pc.check_session(pkg, ectx, oracle, 176, 144, 5, n_streams=2, n_frames=3)
        pc.check_session(pkg, dctx, oracle, 176, 144, 5, n_streams=2, n_frames=3)
        w, h = 320, 240
        enc = pkg.EncoderSession(ectx, w, h, 5, 1)
        dec = pkg.DecoderSession(dctx, w, h, np.stack(pkg.qtables_from_quality(5)[:4]), 1)
        nb, fb = enc.total_blocks, enc.frame_bytes
        d_f, d_c, d_m, d_h = ectx.alloc(fb), ectx.alloc(nb * 512), ectx.alloc(nb * 2), ectx.alloc(nb)
        ev = ectx.event()
        for t in range(4):
            ectx.synth_frames_dev(w, h, [pkg.synth.SEED], t, d_f)
            if t == 0:
                enc.encode_iframe_dev(d_f, d_c)
            else:
                enc.encode_pframe_dev(d_f, d_m, d_h, d_c)
            ectx.record(ev)
            dctx.wait_event(ev)
            if t == 0:
                dec.decode_iframe_dev(d_c)
            else:
                dec.decode_pframe_dev(d_m, d_h, d_c)
            dctx.sync()                       # the next encode overwrites the buffers the decoder reads
            assert np.array_equal(enc.prev_frame(), dec.framebuffer()), t
        ectx.event_destroy(ev)
        enc.close(); dec.close()


def test_blit_dev(pkg, gpu_ctx, oracle):
    """VideoPlane::blit (src/plane.rs:20-29) on device planes vs the oracle's pfvo_blit: random + corner rectangles"""
    assert pc.check_blit_dev(pkg, gpu_ctx, oracle) >= 200


def test_stream_encoder_decoder_vs_oracle(pkg, gpu_ctx, oracle):
    """Encoder -> .pfv bytes -> Decoder; stream bytes and decoded frames equal the oracle's"""
    data = sc.check_stream_roundtrip(pkg, gpu_ctx, oracle, 176, 144, 5, n_frames=7, gop=3, drop_at=(4,))
    sc.check_advance_delta(pkg, gpu_ctx, oracle, data, kinds=[True, True, True, True, False, True, True])
    sc.check_header_errors(pkg, gpu_ctx, data)
    sc.check_stream_roundtrip(pkg, gpu_ctx, oracle, 100, 60, 2, n_frames=3, gop=15)
    sc.check_stream_roundtrip(pkg, gpu_ctx, oracle, 64, 48, 10, n_frames=3, gop=15)
    sc.check_encoder_keeps_nothing(pkg, gpu_ctx)


@pytest.mark.parametrize("geom", [(176, 144, 5, 5, 3), (34, 18, 2, 3, 3), (100, 60, 8, 4, 4)])
def test_corrupted_streams_match_oracle(pkg, gpu_ctx, oracle, geom):
    """byte-flipped / truncated .pfv streams: same frames and same error codes as the oracle's decoder, call by call"""
    w, h, q, n, gop = geom
    data, _ = sc.encode_clip(pkg, gpu_ctx, oracle, w, h, 30, q, n_frames=n, gop=gop)
    stats = sc.check_corrupted_streams(pkg, gpu_ctx, oracle, data, n_trials=120, seed=w + h)
    assert stats["trials"] == 120 and stats["frames"] > 0
    sc.check_lookahead_reset(pkg, gpu_ctx, data, n_frames=n)


@pytest.mark.parametrize("geom", [(48, 32, 2), (176, 144, 3), (34, 18, 1), (640, 360, 2)])
def test_device_entropy_payloads(pkg, gpu_ctx, oracle, geom):
    """k_ent_* payload bytes == the oracle's write_iframe_packet / write_pframe_packet restatement"""
    w, h, S = geom
    assert pc.check_device_entropy(pkg, gpu_ctx, oracle, w, h, n_streams=S, seed=w) == 10 * S


def test_device_entropy_1080p(pkg, gpu_ctx, oracle):
    assert pc.check_device_entropy(pkg, gpu_ctx, oracle, 1920, 1080, n_streams=2, seed=9, kinds=("typical", "edges")) == 8


def test_device_entropy_4k(pkg, gpu_ctx, oracle):
    """48 960 macroblocks = 765 scan workgroups: the group prefix of k_ent_codes takes more than one 256-wide chunk"""
    assert pc.check_device_entropy(pkg, gpu_ctx, oracle, 3840, 2160, n_streams=1, seed=10, kinds=("typical",)) == 2


def test_stream_encoder_host_and_device_entropy_agree(pkg, gpu_ctx):
    """Encoder(device_entropy=False) and Encoder(device_entropy=True) write the same .pfv bytes"""
    import io
    w, h = 320, 240
    st = pkg.SyntheticStream(w, h)
    outs = []
    for dev in (False, True):
        buf = io.BytesIO()
        enc = pkg.Encoder(buf, w, h, 30, 6, gpu_ctx, device_entropy=dev)
        for t in range(5):
            fr = pkg.VideoFrame.from_packed(w, h, st.frame(t)) if hasattr(pkg.VideoFrame, "from_packed") else sc.frame_of(pkg, w, h, st.frame(t))
            (enc.encode_iframe if t % 3 == 0 else enc.encode_pframe)(fr)
        enc.finish()
        enc.close()
        outs.append(buf.getvalue())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


def test_async_entropy_stream(pkg, gpu_ctx, oracle):
    pc.check_async_entropy(pkg, gpu_ctx, oracle, 320, 240, n_streams=3)
    pc.check_async_entropy(pkg, gpu_ctx, oracle, 1920, 1080, n_streams=2, n_frames=4)


def test_colour_conversions_exhaustive(pkg, gpu_ctx, oracle):
    """every RGB colour and every (Y, U, V) triple through the device converters vs the oracle (src/lib.rs:337-394)"""
    pc.check_colour_conversions(pkg, gpu_ctx, oracle, exhaustive=True)


@pytest.mark.parametrize("geom", [(65534, 16), (16, 65534), (65534, 2), (2, 65534)])
def test_extreme_aspect_ratios(pkg, gpu_ctx, oracle, geom):
    """the widest / tallest frames the container can describe (u16 dimensions, src/enc.rs:195-196)"""
    w, h = geom
    stats = pc.check_session(pkg, gpu_ctx, oracle, w, h, 5, n_streams=1, n_frames=2, gop=15, threads=os.cpu_count() or 1)
    assert stats["mbs"] == pkg._lib.load().pfv_total_blocks(w, h) and 0 <= stats["coded"] <= stats["mbs"]     # one p-frame, every macroblock accounted for
    assert pc.check_device_entropy(pkg, gpu_ctx, oracle, w, h, n_streams=1, seed=w + h, kinds=("typical",)) == 2


def test_batch_encoder(pkg, gpu_ctx, oracle):
    sc.check_batch_encoder(pkg, gpu_ctx, oracle, 176, 144, 5, n_streams=5, n_frames=5, gop=3)
    sc.check_batch_encoder(pkg, gpu_ctx, oracle, 640, 360, 8, n_streams=2, n_frames=3, gop=15)


def test_batch_encoder_writer_failure_is_reported(pkg, gpu_ctx):
    sc.check_batch_encoder_writer_failure(pkg, gpu_ctx)


def test_batch_decoder(pkg, gpu_ctx, oracle):
    sc.check_batch_decoder(pkg, gpu_ctx, oracle, 176, 144, 5, n_streams=5, n_frames=5, gop=3)
    sc.check_batch_decoder(pkg, gpu_ctx, oracle, 640, 360, 7, n_streams=2, n_frames=3, gop=15)
    sc.check_batch_decoder(pkg, gpu_ctx, oracle, 64, 48, 10, n_streams=3, n_frames=3, gop=2, noise=True)


def test_lists_decode(pkg, gpu_ctx):
    """coefficient lists (round 5) through pfv_dec_*_lists_dev == the dense decode, ragged and whole strips, up to 960 x 540"""
    assert pc.check_lists_decode(pkg, gpu_ctx, 100, 60, n_streams=2) > 0
    assert pc.check_lists_decode(pkg, gpu_ctx, 640, 360, n_streams=3, seed=12) > 0
    assert pc.check_lists_decode(pkg, gpu_ctx, 960, 540, n_streams=1, seed=15, kinds=("dense", "dense", "typical", "edges")) > 0


def test_sparse_decode(pkg, gpu_ctx):
    pc.check_sparse_decode(pkg, gpu_ctx, 100, 60, n_streams=2)
    pc.check_sparse_decode(pkg, gpu_ctx, 640, 360, n_streams=3, seed=12)


def test_colour_utils(pkg, gpu_ctx, oracle):
    pc.check_colour_utils(pkg, gpu_ctx, oracle)


def test_misaligned_device_frames(pkg, gpu_ctx, oracle):
    pc.check_misaligned_device_frames(pkg, gpu_ctx, oracle)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_plane_operators(pkg, gpu_ctx, oracle, seed):
    pc.fuzz_plane_ops(pkg, gpu_ctx, oracle, n_cases=40, seed=seed)


def test_gop_objects_patterns(pkg, gpu_ctx, oracle):
    """pfv_gop_encoder / pfv_gop_decoder on packet patterns with drop frames, leading p-frames, unequal GOPs, at batch shapes that cut
    runs in the middle: same .pfv bytes as the serial Encoder and the oracle, same frames / results call by call"""
    sc.check_gop_objects(pkg, gpu_ctx, oracle, 64, 48, 5, "IPPPIPDPPIPPP", shapes=((8, 15), (2, 15), (2, 3), (1, 2)))
    sc.check_gop_objects(pkg, gpu_ctx, oracle, 50, 38, 3, "PPIPIPPPPDIP", shapes=((3, 15), (2, 2)))
    sc.check_gop_objects(pkg, gpu_ctx, oracle, 640, 360, 8, "IPPPPPPIPPPPPPIPP" + "D" + "IPPPP", shapes=((4, 7), (3, 4)), threads=8, dec_threads=4)
    sc.check_gop_encoder_flush_and_errors(pkg, gpu_ctx, oracle)


@pytest.mark.parametrize("geom", [(3840, 2160, 31, 8), (1920, 1080, 61, 3)])
def test_gop_objects_config4_vs_serial_and_oracle(pkg, gpu_ctx, oracle, geom):
    """BASELINE config #4's stream through the GOP-batched objects: 4K x 31 frames (i-frames at 0, 15, 30) in one batch of 3 groups and
    1080p x 61 frames (5 GOPs) in batches of 3 groups: .pfv bytes == the serial product Encoder's == the oracle's, every decoded frame
    == the oracle's decoder, in order."""
    w, h, n, max_gops = geom
    fb = int(pkg._lib.load().pfv_frame_bytes(w, h))
    dev = gpu_ctx.alloc(fb)
    frames = []
    for t in range(n):
        gpu_ctx.synth_frames_dev(w, h, [pkg.synth.SEED], t, dev)
        a = np.empty(fb, np.uint8)
        gpu_ctx.download(a, dev)
        frames.append(a)
    gpu_ctx.free(dev)
    pattern = "".join("I" if t % 15 == 0 else "P" for t in range(n))
    threads = min(32, len(os.sched_getaffinity(0)))
    data = sc.check_gop_objects(pkg, gpu_ctx, oracle, w, h, 5, pattern, shapes=((max_gops, 15),), frame_src=lambda t: frames[t], threads=threads, dec_threads=8)
    oracle.L.pfvo_pool_shutdown()
    assert len(data) > 1000


@pytest.mark.parametrize("geom", [(48, 32), (130, 70), (320, 240)])
def test_gop_decoder_corrupted_streams(pkg, gpu_ctx, oracle, geom):
    w, h = geom
    data, _ = sc.encode_pattern(pkg, gpu_ctx, oracle, w, h, 5, "IPPIPPPIPDPIP", lambda buf: pkg.Encoder(buf, w, h, 30, 5, gpu_ctx), with_oracle=False)
    stats = sc.check_gop_decoder_corrupted(pkg, gpu_ctx, oracle, data, n_trials=60, seed=w + h)
    assert stats["trials"] == 60 and stats["errors"] > 10 and stats["frames_after_an_error"] > 0


@pytest.mark.parametrize("geom", [(96, 64), (640, 360), (1920, 1080)])
def test_gop_decoder_device_entropy(pkg, gpu_ctx, oracle, geom):
    """k_entd_*: packet payloads read by the self-synchronising device stage (PFV_OPT_ENTROPY_DECODE): every packet of the synthetic
    content on the device, unsettled / periodic / long-code content through whichever side, always the oracle's frames"""
    out = sc.check_gop_device_entropy(pkg, gpu_ctx, oracle, *geom)
    assert out["noise"]["packets_read_on_device"] >= 1, out
    oracle.L.pfvo_pool_shutdown()


def test_one_symbol_table_lists(pkg, gpu_ctx, oracle):
    """a degenerate code table whose values outnumber what the list pool set aside for the packet's bits: host parser, second parse, spill buffer"""
    assert sc.check_one_symbol_table_lists(pkg, gpu_ctx, oracle) == 5
    assert sc.check_one_symbol_table_lists(pkg, gpu_ctx, oracle, 320, 240, seed=6) == 5


def test_device_block_headers(pkg, gpu_ctx, oracle):
    """k_hdr_*: the p-frames' block headers read on the device, 1080p (12 240 macroblocks: up to 96 header workgroups) and a ragged geometry"""
    sc.check_device_block_headers(pkg, gpu_ctx, oracle, 1920, 1080, pattern="IPPP")
    sc.check_device_block_headers(pkg, gpu_ctx, oracle, 1000, 562, pattern="IPPPPP", seed=9)
    oracle.L.pfvo_pool_shutdown()


def test_gop_decoder_dense_iframe_failure(pkg, gpu_ctx, oracle):
    assert sc.check_gop_decoder_dense_iframe_failure(pkg, gpu_ctx, oracle) >= 1
    sc.check_gop_decoder_dense_iframe_failure(pkg, gpu_ctx, oracle, 320, 240, 0, require_hit=False)   # another geometry; a flip may leave the packet parseable


def test_config4_4k_gop15_stream_vs_oracle(pkg, gpu_ctx, oracle):
    """BASELINE config #4 at its stated geometry and GOP pattern, under the driver's eyes: 3840x2160, 31 frames (i-frames at
    0, 15 and 30 -> two full GOP boundaries, README.md:34-41), quality 5, product Encoder -> .pfv bytes -> product Decoder
    (src/dec.rs:169-224 loop) against OracleStreamEncoder / OracleStreamDecoder: the stream bytes and every decoded frame.
    Frames come from the device-side generator (one of them is also checked against synth.py here)."""
    W, H, N = 3840, 2160, 31
    fb = int(pkg._lib.load().pfv_frame_bytes(W, H))
    dev = gpu_ctx.alloc(fb)
    frames = []
    for t in range(N):
        gpu_ctx.synth_frames_dev(W, H, [pkg.synth.SEED], t, dev)
        a = np.empty(fb, np.uint8)
        gpu_ctx.download(a, dev)
        frames.append(a)
    gpu_ctx.free(dev)
    assert np.array_equal(frames[16], pkg.SyntheticStream(W, H).frame(16))
    threads = min(32, len(os.sched_getaffinity(0)))
    data = sc.check_stream_roundtrip(pkg, gpu_ctx, oracle, W, H, 5, n_frames=N, gop=15, frame_src=lambda t: frames[t], threads=threads)
    oracle.L.pfvo_pool_shutdown()
    # packet walk: types 1, 2 x 14, 1, 2 x 14, 1, then EOF
    pos, kinds = 20 + 4 * 128, []
    while pos < len(data):
        kinds.append(data[pos])
        pos += 5 + int.from_bytes(data[pos + 1:pos + 5], "little")
    assert kinds == ([1] + [2] * 14) * 2 + [1, 0]
