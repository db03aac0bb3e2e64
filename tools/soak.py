#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (not part of the test suite): many more geometries / contents / qualities than
tests/test_gpu_parity.py runs, same checkers, product vs oracle.  usage: python tools/soak.py [minutes]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def iteration(pkg, ctx, oracle, s, it, stats, small=False):
    """one randomised pass over every checker (seed s).  small: geometries the CPU emulator finishes in seconds (tests/test_emulated_kernels.py
    runs a few of these passes; the GPU soak runs the full sizes)"""
    import parity_cases as pc
    import stream_cases as sc
    r = np.random.default_rng(s)
    # round 3: every iteration under a random lane mapping / compaction setting / content kind
    L = pkg._lib
    lanes = [L.PFV_LANES_AUTO, L.PFV_LANES_PER_MB_8, L.PFV_LANES_PER_MB_16][int(r.integers(0, 3))]
    ctx.set_option(L.PFV_OPT_LANE_MAPPING, lanes)
    ctx.set_option(L.PFV_OPT_TILE_COMPACTION, int((0, 1, 1, 2)[int(r.integers(0, 4))]))      # strips / tile compaction (default) / split kernels
    ctx.set_option(L.PFV_OPT_ENC_TRANSFORM, L.PFV_ENC_TRANSFORM_INT if int(r.integers(0, 6)) == 0 else L.PFV_ENC_TRANSFORM_AUTO)
    kind = ["pan", "low_motion", "static"][int(r.integers(0, 3))]
    stats["lanes8"] = stats.get("lanes8", 0) + (lanes == L.PFV_LANES_PER_MB_8)
    n_plane = 6 if small else 25
    pc.fuzz_plane_ops(pkg, ctx, oracle, n_cases=n_plane, seed=s, max_w=120 if small else 700, max_h=64 if small else 300)
    stats["plane_cases"] += n_plane
    w, h = (2 * int(r.integers(1, 40)), 2 * int(r.integers(1, 28))) if small else (2 * int(r.integers(1, 200)), 2 * int(r.integers(1, 120)))
    q = int(r.integers(0, 11))
    S = int(r.integers(1, 4))
    pc.check_session(pkg, ctx, oracle, w, h, q, n_streams=S, n_frames=int(r.integers(2, 6)), gop=int(r.integers(1, 5)), kind=kind)
    stats["sparse_tile_cases"] = stats.get("sparse_tile_cases", 0) + pc.check_sparse_coded_tiles(pkg, ctx, oracle, seed=s, sizes=((2 * int(r.integers(20, 70)), 2 * int(r.integers(10, 40))) if small else (2 * int(r.integers(40, 300)), 2 * int(r.integers(20, 120))),))
    stats["sessions"] += 1
    stats["entropy_payloads"] += pc.check_device_entropy(pkg, ctx, oracle, w, h, n_streams=S, seed=s)
    nf, gop = int(r.integers(2, 7)), int(r.integers(1, 4))
    data = sc.check_stream_roundtrip(pkg, ctx, oracle, w, h, q, n_frames=nf, gop=gop)
    stats["stream_roundtrips"] += 1
    stats["corrupted_trials"] += sc.check_corrupted_streams(pkg, ctx, oracle, data, n_trials=12 if small else 40, seed=s)["trials"]
    sc.check_batch_encoder(pkg, ctx, oracle, w, h, q, n_streams=S, n_frames=3, gop=2)
    sc.check_batch_decoder(pkg, ctx, oracle, w, h, q, n_streams=S, n_frames=4, gop=3)
    stats["batch"] += 1
    pc.check_sparse_decode(pkg, ctx, w, h, n_streams=S, seed=s)
    # round 4: the GOP-batched paths -- the slots of one session / one batch hold the GOPs of ONE stream
    g_gop, g_frames = int(r.integers(1, 6)), int(r.integers(2, 14))
    pc.check_gop_batched_session(pkg, ctx, oracle, w, h, q, n_frames=g_frames, gop=g_gop, seed=s, kind=kind)
    pattern = "".join("IPPPD"[int(k)] for k in r.integers(0, 5, int(r.integers(3, 14))))     # any packet order the API allows, leading p-frames included
    shapes = ((int(r.integers(1, 6)), int(r.integers(1, 6))), (8, 15))
    gdata = sc.check_gop_objects(pkg, ctx, oracle, w, h, q, pattern, shapes=shapes, dec_threads=int(r.integers(0, 4)))
    stats["gop_batched"] = stats.get("gop_batched", 0) + 1
    if pattern.count("I") + pattern.count("P") >= 2:
        stats["gop_corrupted_trials"] = stats.get("gop_corrupted_trials", 0) + sc.check_gop_decoder_corrupted(
            pkg, ctx, oracle, gdata, n_trials=4 if small else 12, seed=s, shapes=shapes)["trials"]
    # round 4: the decoder's entropy stage on the device (k_entd_*): unsettled / periodic / long-code content on valid streams, and the
    # counters of every GOP-batched decoder the checks above ran (half of them read their payloads on the device)
    if it % 3 == 0:
        ed = sc.check_gop_device_entropy(pkg, ctx, oracle, w, h, quality=q, pattern="".join("IPPP"[int(k)] for k in r.integers(0, 4, int(r.integers(2, 9)))),
                                         min_device_share=0.0, expect_unsettled=False)
        stats["device_entropy_cases"] = stats.get("device_entropy_cases", 0) + 1
        stats["device_entropy_noise_on_device"] = stats.get("device_entropy_noise_on_device", 0) + ed["noise"]["packets_read_on_device"]
    # round 5: coefficient lists into the decode kernels, the p-frames' block headers read on the device (geometries large enough for
    # several header workgroups and several workgroups of the run-stream read), the whole-clip form of the GOP-batched session check
    stats["list_entries"] = stats.get("list_entries", 0) + pc.check_lists_decode(pkg, ctx, w, h, n_streams=S, seed=s)
    if it % 2 == 0:
        bw, bh = (2 * int(r.integers(60, 140)), 2 * int(r.integers(30, 72))) if small else (2 * int(r.integers(120, 640)), 2 * int(r.integers(60, 360)))
        sc.check_device_block_headers(pkg, ctx, oracle, bw, bh, quality=q, pattern="I" + "P" * int(r.integers(1, 4)), seed=int(r.integers(1, 1000)))
        stats["device_header_clips"] = stats.get("device_header_clips", 0) + 1
        cg = int(r.integers(2, 6))
        pc.check_gop_batched_clip(pkg, ctx, oracle, w, h, q, n_frames=int(r.integers(cg + 1, 4 * cg)), gop=cg, seed=s, dec_gops=int(r.integers(1, 5)))
        stats["gop_batched_clips"] = stats.get("gop_batched_clips", 0) + 1
    stats.update({"gop_dec_" + k: v for k, v in sc.ENTROPY_COUNTS.items()})
    if w % 32 == 0:     # the fused retframe crop needs 16-byte rows in every plane
        pc.check_gop_graph(pkg, ctx, oracle, w, h, n_streams=S, n_frames=int(r.integers(2, 5)), quality=q)
        stats["graphs"] = stats.get("graphs", 0) + 1


def main():
    import __graft_entry__ as g
    g.build()
    pkg = g.load_package()
    __import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
    from oracle_bind import Oracle
    budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 300.0
    oracle = Oracle()
    stats = {"plane_cases": 0, "sessions": 0, "entropy_payloads": 0, "corrupted_trials": 0, "stream_roundtrips": 0, "batch": 0}
    t_end = time.time() + budget
    rng = np.random.default_rng(int(time.time()))
    seed0 = int(rng.integers(1 << 30))
    print("seed", seed0, flush=True)
    it = 0
    with pkg.Context(0) as ctx:
        while time.time() < t_end:
            iteration(pkg, ctx, oracle, seed0 + it, it, stats)
            it += 1
            if it % 10 == 0:
                print(it, json.dumps(stats), flush=True)
    print("OK", json.dumps(stats))


if __name__ == "__main__":
    main()
