#!/usr/bin/env python
"""bench.py -- macroblocks/s (encode+decode) on synthetic 1080p YUV 4:2:0, MI355X.

One "step" = one GOP-15 (1 i-frame + 14 p-frames, README.md:34-41 pattern) of S independent
synthetic 1080p streams per GPU, each frame ENCODED (with closed-loop reconstruction) and then
DECODED from the coefficients just produced, all streams batched into one kernel launch per
frame operation.  Inputs (the raw frames) are resident in HBM before the timed region starts;
coefficients / motion vectors / reconstructed frames never leave HBM.  Entropy coding (host)
is outside this path.

    python bench.py --gpus N --steps K --warmup W [--streams S] [--width 1920 --height 1080]

N > 1: one process per GPU (torchrun); the streams are independent, so they are sharded
across ranks with NO data-path collective ("weak" scaling: S streams per GPU).  RCCL is used
only for the stream-assignment broadcast and the final counter gather.

The JSON line also carries
  roofline     -- dominant kernel (k_enc_pframe): algorithmic bytes per launch (1284 B per
                  macroblock) / its average HIP-event duration on the
                  context's own stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm with its fork/join
                  structure; the Rust reference itself cannot be built here) timed on this
                  node's host cores on one GOP of one stream of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOP = 15
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_MB_PENC = 1284        # src 256 + ref 256 + coef 512 + mv/flag 4 + recon 256
# algorithmic bytes per macroblock of the other three codec kernels + 256 for the retframe crop the
# decode kernels fuse (src/dec.rs:195-197, 209-211)
BYTES_PER_MB = {"k_enc_iframe": 1024, "k_enc_pframe": BYTES_PER_MB_PENC, "k_dec_iframe": 768 + 256, "k_dec_pframe": 1028 + 256}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=96, help="independent streams per GPU, batched per launch")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--quality", type=int, default=5)
    ap.add_argument("--unique", type=int, default=2, help="distinct synthetic streams generated on the host per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-entropy", action="store_true", help="skip the extra encode_to_payload measurement (device entropy stage)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the decoder == encoder check (ablation builds of the kernels produce invalid results by construction)")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="skip the two-stream variant of that measurement (profiling runs: keeps per-kernel durations free of time-slicing)")
    return ap.parse_args()


def cpu_baseline(pkg, width, height, quality, frames_one_stream):
    """encode+decode GOPs of one stream with the CPU oracle on the host cores.  The reference sizes its rayon
    pool from a caller-chosen num_threads (src/enc.rs:54); several pool sizes are tried on one GOP each and the
    best one is then timed for a few more GOPs, so the baseline is not handicapped by a bad thread count."""
    from oracle_bind import Oracle, OracleDecoder
    ora = Oracle()
    ncpu = os.cpu_count() or 1
    tabs = np.stack(ora.qtables(quality)[:4])

    penc = {"s": 0.0, "n": 0}

    def run(threads, max_reps, budget_s, record=False):
        ora.L.pfvo_pool_shutdown()          # fresh pool of exactly `threads` workers
        enc = ora.encoder(width, height, quality, threads=threads)
        dec = OracleDecoder(ora, width, height, tabs, threads=threads)
        t0 = time.perf_counter()
        reps = 0
        while True:
            for t, f in enumerate(frames_one_stream):
                if t == 0:
                    dec.decode_iframe(enc.encode_iframe(f))
                else:
                    t1 = time.perf_counter()
                    r = enc.encode_pframe(f)
                    if record:
                        penc["s"] += time.perf_counter() - t1
                        penc["n"] += enc.total_blocks
                    dec.decode_pframe(*r)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= max_reps:
                break
        assert np.array_equal(dec.framebuffer(), enc.prev_frame())
        return reps * len(frames_one_stream) * enc.total_blocks / el, reps, el

    trials = {}
    for th in sorted({1, min(8, ncpu), min(32, ncpu), ncpu}):
        trials[th] = run(th, 1, 5.0)[0]
    best = max(trials, key=trials.get)
    rate, reps, el = run(best, 40, 3.0, record=True)     # ~3 s of wall time on the best pool size
    n_mb = reps * len(frames_one_stream) * pkg._lib.load().pfv_total_blocks(width, height)
    return {"value": rate, "unit": "macroblocks/s", "cores": best, "kind": "port",
            "pframe_encode_value": penc["n"] / penc["s"] if penc["s"] > 0 else None,
            "sample": f"{reps} x GOP-{len(frames_one_stream)} encode+decode of one {width}x{height} stream ({n_mb} macroblocks, "
                      f"{el:.1f} s); C oracle = port of the reference's algorithm with its per-plane fork/join, persistent "
                      f"pool of {best} threads (best of {dict((k, round(v)) for k, v in trials.items())} on {ncpu} host CPUs)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # developer dry-run of the N > 1 control flow on a one-GPU box: PFV_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # uses gloo (RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("PFV_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as graft
    graft.build_hip()
    pkg = graft.load_package()
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")

    W, H, S, Q = args.width, args.height, args.streams, args.quality

    # ---- stream assignment: rank 0 decides, everyone learns it through one tiny broadcast
    # (the only "scatter" this path has: stream ids / seeds, a few hundred bytes)
    table = shard.assign_streams(n_streams_total=S * world, world=world, base_seed=pkg.synth.SEED)
    if world > 1:
        t = torch.tensor(table if rank == 0 else np.zeros_like(table), device=dev)
        dist.broadcast(t, src=0)
        table = t.cpu().numpy()
    mine = shard.streams_of_rank(table, rank)
    assert len(mine) == S

    # ---- synthetic input, resident in HBM: [GOP][S][frame_bytes]
    uniq = max(1, min(args.unique, S))
    fb = int(pkg._lib.load().pfv_frame_bytes(W, H))
    host = np.empty((GOP, uniq, fb), dtype=np.uint8)
    for u in range(uniq):
        st = pkg.SyntheticStream(W, H, seed=int(mine[u][1]))
        for t in range(GOP):
            host[t, u] = st.frame(t)
    frames = torch.from_numpy(host).to(dev)                              # [GOP, uniq, fb]
    frames = frames[:, torch.arange(S, device=dev) % uniq].contiguous()  # [GOP, S, fb]

    ctx = pkg.Context(local_rank)
    enc = pkg.EncoderSession(ctx, W, H, Q, S)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), S)
    n_mb = enc.total_blocks
    coef = torch.empty((S, n_mb, 256), dtype=torch.int16, device=dev)
    mv = torch.empty((S, n_mb, 2), dtype=torch.int8, device=dev)
    has = torch.empty((S, n_mb), dtype=torch.uint8, device=dev)
    out_frames = torch.empty((S, fb), dtype=torch.uint8, device=dev)
    dec.set_output_dev(out_frames.data_ptr())                            # retframe crop (src/dec.rs:209-211) fused into decode
    stream = torch.cuda.ExternalStream(ctx.stream, device=dev)           # HIP events on the kernels' own stream
    torch.cuda.synchronize()

    ev_pairs = []
    ev_other = {"k_enc_iframe": [], "k_dec_iframe": [], "k_dec_pframe": []}

    def step(timed: bool):
        # HIP events on the kernels' own stream bracket every launch of the timed steps: k_enc_pframe for the roofline
        # object, the other three for the per-kernel table (one event = ~1 us of host time, inside the timed region)
        def ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e
        for t in range(GOP):
            f = frames[t].data_ptr()
            if t == 0:
                a = ev() if timed else None
                enc.encode_iframe_dev(f, coef.data_ptr())
                b = ev() if timed else None
                dec.decode_iframe_dev(coef.data_ptr())
                if timed:
                    ev_other["k_enc_iframe"].append((a, b))
                    ev_other["k_dec_iframe"].append((b, ev()))
            else:
                a = ev() if timed else None
                enc.encode_pframe_dev(f, mv.data_ptr(), has.data_ptr(), coef.data_ptr())
                b = ev() if timed else None
                dec.decode_pframe_dev(mv.data_ptr(), has.data_ptr(), coef.data_ptr())
                if timed:
                    ev_pairs.append((a, b))
                    ev_other["k_dec_pframe"].append((b, ev()))

    for _ in range(args.warmup):
        step(False)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if not args.no_verify:
        dec.check()

    # sanity inside the bench: decoder output == encoder reconstruction, and the p-frames did real work
    assert args.no_verify or np.array_equal(enc.prev_frame(), dec.framebuffer()), "decoder framebuffer != encoder reconstruction"
    coded_frac = float(has.float().mean().item())

    pe_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs])) if ev_pairs else float("nan")

    # ---- beside the headline (never part of `value`): the encoder alone with its entropy stage on the device, i.e.
    # frames in HBM -> packet payloads in HBM (k_enc_* + k_ent_*), one GOP per pass
    ent = None
    if not args.no_entropy:
        # two sets of encode outputs: with the stage on its own HIP stream the k_ent_* kernels of frame t (memory-bound)
        # overlap k_enc_pframe of frame t+1 (VALU-bound); pack(t+1) orders later main-stream work behind pack(t)'s reads
        sets = [(coef, mv, has), (torch.empty_like(coef), torch.empty_like(mv), torch.empty_like(has))]

        def encode_gop():
            for t in range(GOP):
                f = frames[t].data_ptr()
                c, m, h = sets[t & 1]
                if t == 0:
                    enc.encode_iframe_dev(f, c.data_ptr())
                    enc.pack_iframe_dev(c.data_ptr())
                else:
                    enc.encode_pframe_dev(f, m.data_ptr(), h.data_ptr(), c.data_ptr())
                    enc.pack_pframe_dev(m.data_ptr(), h.data_ptr(), c.data_ptr())

        def measure(async_stream):
            enc.enable_entropy(async_stream=async_stream)
            encode_gop()
            enc.entropy_join()
            ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(1, min(args.steps, 5))
            e0.record(stream)
            for _ in range(reps):
                encode_gop()
            enc.entropy_join()          # the kernels' stream waits for the entropy stream before the closing event
            e1.record(stream)
            ctx.sync()
            return e0.elapsed_time(e1) / reps, enc.payload_sizes()

        serial_ms, sizes = measure(False)
        gop_ms = None
        if not args.no_two_stream:
            gop_ms, sizes_b = measure(True)
            assert np.array_equal(sizes, sizes_b), "entropy stage: two-stream and same-stream runs disagree"
        ent = {"value": GOP * S * n_mb / (serial_ms * 1e-3), "unit": "macroblocks/s", "ms_per_gop": serial_ms,
               "two_stream_value": GOP * S * n_mb / (gop_ms * 1e-3) if gop_ms else None, "two_stream_ms_per_gop": gop_ms,
               "last_pframe_payload_bytes_per_stream": float(np.mean(sizes)),
               "note": "encode only, frames in HBM -> .pfv packet payloads in HBM: k_enc_iframe/k_enc_pframe + the device "
                       "entropy stage (k_ent_scan/codes/init/pack), HIP-event time over whole GOPs; two_stream_*: the stage "
                       "on a second HIP stream with double-buffered encode outputs (k_enc_pframe's 5 wavefronts/SIMD fill "
                       "the VGPR file, so the kernels time-slice instead of co-residing: no gain expected)"}

    elt = torch.tensor([el], device=dev, dtype=torch.float64)
    cnt = torch.tensor([float(args.steps) * GOP * S * n_mb], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)       # max over ranks
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)       # counter gather
    el_max, total_mb = float(elt.item()), float(cnt.item())

    if rank == 0:
        launch_mbs = S * n_mb
        achieved = launch_mbs * BYTES_PER_MB_PENC / (pe_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
        # tools/gpu_pmc.sh); only quoted when it was collected on this exact configuration
        traffic, valu = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            c = pm["config"]
            if (int(c["streams"]), int(c["width"]), int(c["height"]), int(c["quality"])) == (S, W, H, Q):
                traffic = pm["kernels"]["k_enc_pframe"]["traffic_bytes"]
                n_valu = pm["kernels"]["k_enc_pframe"].get("valu_wave_instructions")
                if n_valu:
                    # what actually bounds the kernel: VALU issue.  1024 SIMDs at the 2.4 GHz maximum clock; a wave64
                    # instruction occupies its SIMD for 2 (plain VOP2) to 4+ (VOP3 / DPP / SDWA / v_dot4) cycles
                    valu = {"wave_instructions_per_launch": n_valu,
                            "simd_cycles_per_instruction": pe_ms * 1e-3 * 2.4e9 * 1024 / n_valu,
                            "note": "SQ_INSTS_VALU from the committed PMC pass (profiles/pmc_traffic.json), this run's launch "
                                    "time, 1024 SIMDs x 2.4 GHz: the kernel retires one wave64 VALU instruction per ~4.5 SIMD "
                                    "cycles, i.e. it is bound by VALU issue, not by the HBM roof quoted in frac"}
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "macroblocks/s (encode+decode) 1080p YUV420",
            "value": total_mb / el_max,
            "unit": "macroblocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": f"synthetic ({uniq} distinct integer-hash texture streams per GPU tiled to {S}; quality {Q})",
            "config": {"workload": f"{W}x{H} YUV420 GOP-{GOP} encode+decode, {S} independent streams per GPU batched per launch",
                       "streams_per_gpu": S, "macroblocks_per_frame": n_mb, "quality": Q,
                       "pframe_coded_fraction": round(coded_frac, 4), "parallelism": f"streams sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "kernel": "k_enc_pframe", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": launch_mbs * BYTES_PER_MB_PENC,
                         "avg_launch_ms": pe_ms, "macroblocks_per_launch": launch_mbs,
                         "algorithmic_bytes_per_macroblock": BYTES_PER_MB_PENC, "valu": valu},
        }
        res["pframe_encode"] = {"value": launch_mbs / (pe_ms * 1e-3), "unit": "macroblocks/s",
                                "note": "k_enc_pframe alone (motion search + residual DCT + closed-loop reconstruction), HIP-event time"}
        kern = {"k_enc_pframe": pe_ms}
        kern.update({k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev_other.items() if v})
        res["kernels"] = {k: {"avg_launch_ms": ms, "macroblocks_per_s": launch_mbs / (ms * 1e-3),
                              "algorithmic_GBps": launch_mbs * BYTES_PER_MB[k] / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": launch_mbs * BYTES_PER_MB[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                          for k, ms in kern.items()}
        if ent:
            res["encode_to_payload"] = ent
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(pkg, W, H, Q, [host[t, 0] for t in range(GOP)])
            if res["cpu_baseline"].get("pframe_encode_value"):
                res["pframe_encode"]["vs_cpu_baseline"] = res["pframe_encode"]["value"] / res["cpu_baseline"]["pframe_encode_value"]
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)

    enc.close()
    dec.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
