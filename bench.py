#!/usr/bin/env python
"""bench.py -- macroblocks/s (encode+decode) on synthetic YUV 4:2:0, MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload gop1080p|config5] [--streams S]

Workloads (BASELINE.json `configs`):
  gop1080p (default; the configuration the metric is quoted on): one step = one GOP-15 (1 i-frame + 14 p-frames,
      README.md:34-41 pattern) of S independent synthetic 1080p streams per GPU, every frame ENCODED (with closed-loop
      reconstruction) and then DECODED from the coefficients just produced, all streams batched into one kernel launch
      per frame operation.  Every stream is distinct (its own seed), generated on the device.
  config5: one 3840x2160, 300-frame, GOP-15 stream per GPU (seed = base + rank), one step = the whole stream encoded +
      decoded, one launch per frame operation (48 720 macroblocks per launch).  At N = 1 this is config #4.
Inputs (the raw frames) are resident in HBM before the timed region starts; coefficients / motion vectors /
reconstructed frames never leave HBM.  Entropy coding is outside `value` (reported beside it).

N > 1: one process per GPU.  `python bench.py --gpus N` launches its own N ranks (one child process per rank, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); when it is already running under torchrun (those variables set,
which is how the driver starts it) it uses them.  Streams are independent (src/enc.rs:12-26: an Encoder shares nothing
with another), so they are sharded across ranks with NO data-path collective ("weak" scaling: fixed work per GPU); RCCL
-- called through the library (pfv_comm_*, librccl.so), NOT through torch.distributed, so an N > 1 rank is the same
torch-free process as the N = 1 run -- carries only the assignment-table broadcast, the barriers and the final counter
reduction; the ncclUniqueId reaches the ranks over a TCP rendezvous on MASTER_ADDR (pretty-fast-video_amd/comm.py).  With
fewer GPUs than ranks (developer dry run on a one-GPU box) every rank uses device 0 and the control plane stays on the
rendezvous sockets (RCCL refuses two ranks on one device); the JSON says so (`rccl_ranks`: 0, backend "tcp").

The JSON line also carries
  roofline     -- dominant kernel (k_enc_pframe): algorithmic bytes per launch (1284 B per macroblock, SURVEY.md
                  section 8d) / its average HIP-event duration on the context's own stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm with its fork/join structure; the Rust reference
                  itself cannot be built here) timed on this node's host cores on GOPs of one stream of the workload;
  extra        -- (N = 1) single-stream figures (S = 1, S = 8; per-frame launches vs one HIP graph per GOP) and BASELINE
                  config #4 (4K x 300 frames) at kernel / +PCIe / end-to-end scope.
"""
from __future__ import annotations

import argparse
import io
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOP = 15
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMDS, GPU_CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMD16, peak engine clock
VALU_CYCLES_PER_INSTR = 4.2     # issue cost of one wave64 VALU instruction of the encoders' mix (profiles/r02_ubench_valu_rates2.txt: 4.1-4.4)
BYTES_PER_MB_PENC = 1284        # src 256 + ref 256 + coef 512 + mv/flag 4 + recon 256
# algorithmic bytes per macroblock of the other three codec kernels + 256 for the retframe crop the
# decode kernels fuse (src/dec.rs:195-197, 209-211)
BYTES_PER_MB = {"k_enc_iframe": 1024, "k_enc_pframe": BYTES_PER_MB_PENC, "k_dec_iframe": 768 + 256, "k_dec_pframe": 1028 + 256}
BYTES_PER_MB_SURVEY = {"k_enc_iframe": 1024, "k_enc_pframe": 1284, "k_dec_iframe": 768, "k_dec_pframe": 1028}     # SURVEY.md section 8d as written
EMU = os.environ.get("PFV_BENCH_EMU") == "1"    # test-only: kernel sources on the CPU emulator, no GPU (tests/test_sharding.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["gop1080p", "config5"], default="gop1080p")
    ap.add_argument("--streams", type=int, default=None, help="independent streams per GPU, batched per launch (gop1080p: 96)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None, help="frames per step (gop1080p: 15; config5: 300)")
    ap.add_argument("--quality", type=int, default=5)
    ap.add_argument("--force-comm", action="store_true",
                    help="N = 1: still create the (1-rank) RCCL communicator and run the table broadcast, the barriers and the counter "
                         "reduction through it -- the code path of an N > 1 rank, on a one-GPU box")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not collect FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU of this very workload under rocprofv3 (three short child runs); "
                         "roofline.traffic / roofline.issue then quote profiles/pmc_traffic.json")
    ap.add_argument("--serial-gops", action="store_true",
                    help="config5: one launch per FRAME operation (48 720 macroblocks per launch at 4K) instead of the GOP-batched default (frame t of every "
                         "GOP of the stream per launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-entropy", action="store_true", help="skip the extra encode_to_payload measurement (device entropy stage)")
    ap.add_argument("--no-extra", action="store_true", help="skip the single-stream and config-4 side measurements")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the decoder == encoder check (ablation builds of the kernels produce invalid results by construction)")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="skip the two-stream variant of the entropy measurement (profiling runs: keeps per-kernel durations free of time-slicing)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: become the launcher (one child process per GPU on this node).  The children are
    polled: as soon as one ends with an error the others are stopped (they would otherwise sit in the rendezvous or in a collective
    waiting for it) and its code is returned."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]     # MASTER_PORT of the job; the rendezvous derives its own port from it and walks on if that is busy (comm.py)
    procs = []
    nonce = os.urandom(12).hex()      # the job's token on the rendezvous (comm.py)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PFV_RDZV_NONCE=nonce)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code)
    for p in live:                    # a rank failed: the rest of the job goes with it
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------- helpers
def host_cpu_facts():
    facts = {"host_cpus": os.cpu_count() or 1, "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
             "cgroup_cpu_quota": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                facts["cgroup_cpu_quota"] = "max" if txt[0] == "max" else round(int(txt[0]) / int(txt[1]), 2)
            else:
                q = int(txt[0])
                facts["cgroup_cpu_quota"] = "max" if q < 0 else round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()), 2)
            break
        except (OSError, ValueError, IndexError):
            continue
    return facts


def usable_cpus(facts):
    """CPUs' worth of time this container may actually burn: min(affinity mask, cgroup quota)"""
    ncpu = facts["affinity_cpus"] or facts["host_cpus"]
    quota = facts["cgroup_cpu_quota"]
    return ncpu if quota in (None, "max") else max(1, min(ncpu, int(round(float(quota)))))


def cpu_baseline(width, height, quality, frames_one_stream, n_mb, budget_s=24.0):
    """encode+decode GOPs of one stream with the CPU oracle on the host cores.  The reference sizes its rayon pool from a
    caller-chosen num_threads (src/enc.rs:54).  Pool sizes tried: 1, usable_cpus and 2 x usable_cpus -- nothing else: a pool far above the
    cgroup quota is CFS-throttled in 100 ms periods and wins or loses a short trial by chance (round 5: 1.18 M vs 1.96 M macroblocks/s on
    two runs of one commit).  A pass runs >= 3 GOPs and >= `budget_s` / 12 seconds; every pool but the single thread is timed THREE times and
    judged by its median pass; the MEDIAN pass of the winner is reported.  `cores` = usable_cpus (the CPUs' worth of time the container owns:
    min(affinity, cgroup quota)), `threads_best` = the pool size that won."""
    from oracle_bind import Oracle, OracleDecoder
    ora = Oracle()
    facts = host_cpu_facts()
    ncpu = facts["affinity_cpus"] or facts["host_cpus"]
    usable = usable_cpus(facts)
    tabs = np.stack(ora.qtables(quality)[:4])
    min_s = budget_s / 12.0

    def run(threads, min_gops=3):
        ora.L.pfvo_pool_shutdown()          # fresh pool of exactly `threads` workers
        enc = ora.encoder(width, height, quality, threads=threads)
        dec = OracleDecoder(ora, width, height, tabs, threads=threads)
        pe_s, pe_n = 0.0, 0
        t0 = time.perf_counter()
        reps = 0
        while True:
            for t, f in enumerate(frames_one_stream):
                if t == 0:
                    dec.decode_iframe(enc.encode_iframe(f))
                else:
                    t1 = time.perf_counter()
                    r = enc.encode_pframe(f)
                    pe_s += time.perf_counter() - t1
                    pe_n += enc.total_blocks
                    dec.decode_pframe(*r)
            reps += 1
            el = time.perf_counter() - t0
            if (reps >= min_gops and el >= min_s) or el > 6.0 * max(min_s, 0.5):
                break
        assert np.array_equal(dec.framebuffer(), enc.prev_frame())
        return {"rate": reps * len(frames_one_stream) * enc.total_blocks / el, "reps": reps, "el": el, "penc": pe_n / pe_s if pe_s > 0 else None}

    pools = sorted({1, usable, min(2 * usable, max(ncpu, usable))})
    # one pass for the single thread (it never wins on a multi-core box and a pass of it takes three times as long); THREE passes for every other
    # pool, each pool judged by its median: a pool above the cgroup quota is CFS-throttled, single passes of it scatter by +-10 % and a winner
    # picked from single passes flips between runs (round 6: 1.50 M on 16 threads vs 1.69 M on 32, one box, two runs)
    by_pool = {th: sorted([run(th) for _ in range(1 if th == 1 and len(pools) > 1 else 3)], key=lambda r: r["rate"]) for th in pools}
    trials = {th: v[len(v) // 2] for th, v in by_pool.items()}
    best = max(trials, key=lambda th: trials[th]["rate"])
    passes = by_pool[best] if len(by_pool[best]) == 3 else sorted(by_pool[best] + [run(best) for _ in range(2)], key=lambda r: r["rate"])
    med = passes[1]
    ora.L.pfvo_pool_shutdown()
    spread = (passes[2]["rate"] - passes[0]["rate"]) / med["rate"]
    return {"value": med["rate"], "unit": "macroblocks/s", "cores": usable, "usable_cpus": usable, "kind": "port",
            "value_best": med["rate"], "threads_best": best, "value_1thread": trials[1]["rate"],
            "trials_threads_to_value": {str(k): round(v["rate"]) for k, v in trials.items()},
            "passes_of_the_winner": [round(p["rate"]) for p in passes], "spread_of_the_three_passes": round(spread, 4), **facts,
            "pframe_encode_value": med["penc"], "pframe_encode_value_1thread": trials[1]["penc"],
            "sample": f"median of 3 passes of {med['reps']} x GOP-{len(frames_one_stream)} encode+decode of one {width}x{height} stream "
                      f"({med['reps'] * len(frames_one_stream) * n_mb} macroblocks, {med['el']:.1f} s per pass) on a pool of threads_best = {best} workers, chosen among "
                      f"pools of {pools} (each judged by the median of its own three passes of >= 3 GOPs; the single thread by one pass); C oracle = port of the reference's algorithm with its per-plane fork/join over a "
                      f"persistent pool; `cores` = usable_cpus = min(affinity mask, cgroup quota) = the CPUs' worth of time this container owns "
                      f"(host_cpus / cgroup_cpu_quota / affinity_cpus: what it sees of the node)"}


class Timer:
    """HIP events on the kernels' own stream (pfv_event_*: hipEventRecord on the context's stream) or wall-clock stamps (CPU
    emulator runs of the control flow)."""

    class _Ev:
        __slots__ = ("h",)

    def __init__(self, ctx, dev):
        import ctypes
        self.ctx, self.ct = ctx, ctypes
        self.pool, self.all = [], []

    def reserve(self, n):
        """events created ahead of the timed region"""
        if EMU:
            return
        for _ in range(n):
            h = self.ct.c_void_p()
            self.ctx.check(self.ctx._lib.pfv_event_create(self.ctx.handle, self.ct.byref(h)))
            self.pool.append(h)
            self.all.append(h)

    def stamp(self):
        if EMU:
            return time.perf_counter()
        if not self.pool:
            self.reserve(64)
        h = self.pool.pop()
        self.ctx._lib.pfv_event_record(h)
        return h

    def ms(self, a, b):
        if EMU:
            return (b - a) * 1e3
        out = self.ct.c_float()
        self.ctx.check(self.ctx._lib.pfv_event_elapsed_ms(a, b, self.ct.byref(out)))
        return float(out.value)

    def close(self):
        for h in self.all:
            self.ctx._lib.pfv_event_destroy(h)
        self.pool, self.all = [], []


class StreamSet:
    """S independent streams of one geometry resident in HBM: `n_frames` synthetic frames per stream (generated on the
    device from (seed, t)), an encoder session and a decoder session S streams wide, and the buffers between them."""

    def __init__(self, pkg, ctx, W, H, Q, seeds, n_frames, fused_crop=True, kind="pan", dec_ctx=None):
        """dec_ctx: a second context (= a second HIP stream) for the decoder, for wall_pipelined(): the decoder works through pass
        k - 1 while the encoder works through pass k -- Encoder and Decoder are independent objects (src/enc.rs:12-26,
        src/dec.rs:15-28), and one stream's launches cover a fraction of the device.  The encode outputs of a GOP are kept (two
        alternating sets of GOP buffers); the two streams meet ONCE per GOP (pfv_ctx_wait_event)."""
        self.pkg, self.ctx, self.W, self.H, self.Q, self.S, self.n_frames = pkg, ctx, W, H, Q, len(seeds), n_frames
        self.kind, self.dec_ctx = kind, dec_ctx
        self.seeds = [int(s) for s in seeds]
        lib = pkg._lib.load()
        self.fb = int(lib.pfv_frame_bytes(W, H))
        S = self.S
        self.enc = pkg.EncoderSession(ctx, W, H, Q, S)
        self.dec = pkg.DecoderSession(dec_ctx or ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), S)
        self.n_mb = self.enc.total_blocks
        self.launch_streams = S                                 # slots per launch
        self._bufs = []
        self.frames = self._alloc(n_frames * S * self.fb)
        self.coef, self.mv, self.has = self._alloc(S * self.n_mb * 512), self._alloc(S * self.n_mb * 2), self._alloc(S * self.n_mb)
        self.ev_enc = self.ev_dec = None
        if dec_ctx is not None:
            one = lambda: (self._alloc(S * self.n_mb * 512), self._alloc(S * self.n_mb * 2), self._alloc(S * self.n_mb))
            self.pass_sets = [[one() for _ in range(min(n_frames, GOP))] for _ in range(2)]
            self.ev_enc = [ctx.event(), ctx.event()]
            self.ev_dec = [dec_ctx.event(), dec_ctx.event()]
        self.out_frames = self._alloc(S * self.fb)
        if fused_crop:
            self.dec.set_output_dev(self.out_frames)       # retframe crop (src/dec.rs:209-211) fused into decode
        for t in range(n_frames):
            ctx.synth_frames_dev(W, H, self.seeds, t, self.frame_ptr(t), kind=kind)
        ctx.sync()

    def _alloc(self, n):
        p = self.ctx.alloc(max(int(n), 16))
        self._bufs.append(p)
        return p

    def frame_ptr(self, t):
        return self.frames + t * self.S * self.fb

    def host_frames_at(self, stream, t):
        a = np.empty(self.fb, np.uint8)
        self.ctx.download(a, self.frame_ptr(t) + stream * self.fb)
        return a

    def host_frames(self, stream, count):
        return [self.host_frames_at(stream, t) for t in range(count)]

    def step(self, gop=GOP, on_launch=None, sample_frames=2 * GOP):
        """encode + decode every resident frame once; i-frame when t % gop == 0 (README.md:34-41).  on_launch: HIP-event
        brackets around the launches of the first `sample_frames` frames of the pass (every launch of the default
        workload; a 2-GOP sample of a 300-frame stream, whose 20-microsecond launches the event calls would otherwise slow)"""
        enc, dec = self.enc, self.dec
        ev = on_launch
        for t in range(self.n_frames):
            on_launch = ev if t < sample_frames else None
            f = self.frame_ptr(t)
            if t % gop == 0:
                a = on_launch and on_launch()
                enc.encode_iframe_dev(f, self.coef)
                b = on_launch and on_launch()
                dec.decode_iframe_dev(self.coef)
                if on_launch:
                    on_launch("k_enc_iframe", a, b)
                    on_launch("k_dec_iframe", b, on_launch())
            else:
                a = on_launch and on_launch()
                enc.encode_pframe_dev(f, self.mv, self.has, self.coef)
                b = on_launch and on_launch()
                dec.decode_pframe_dev(self.mv, self.has, self.coef)
                if on_launch:
                    on_launch("k_enc_pframe", a, b)
                    on_launch("k_dec_pframe", b, on_launch())

    def encode_pack_pass(self, sets):
        """encode every resident frame and build its packet payload on the device (entropy_side); sets: two alternating sets of
        encode output buffers"""
        enc = self.enc
        for t in range(self.n_frames):
            f = self.frame_ptr(t)
            c, m, h = sets[t & 1]
            if t % GOP == 0:
                enc.encode_iframe_dev(f, c)
                enc.pack_iframe_dev(c)
            else:
                enc.encode_pframe_dev(f, m, h, c)
                enc.pack_pframe_dev(m, h, c)

    def wall_pipelined(self, reps, gop=GOP):
        """macroblocks/s of `reps` passes with the decoder ONE GOP behind the encoder on its own stream (host clock, both streams
        synchronised on both sides); every GOP is encoded AND decoded inside the timed region.  The encode outputs of a GOP go to one
        of two alternating sets of `gop` buffers; the streams meet once per GOP."""
        enc, dec, ectx, dctx = self.enc, self.dec, self.ctx, self.dec_ctx
        gops = [(t0, min(gop, self.n_frames - t0)) for t0 in range(0, self.n_frames, gop)]      # (first frame, frames) of a pass

        def enc_gop(j, t0, n):
            for t in range(t0, t0 + n):
                coef, mv, has = self.pass_sets[j & 1][t - t0]
                if t == t0:
                    enc.encode_iframe_dev(self.frame_ptr(t), coef)
                else:
                    enc.encode_pframe_dev(self.frame_ptr(t), mv, has, coef)
            ectx.record(self.ev_enc[j & 1])

        def dec_gop(j, t0, n):
            dctx.wait_event(self.ev_enc[j & 1])                  # the GOP's encode outputs are complete
            for t in range(t0, t0 + n):
                coef, mv, has = self.pass_sets[j & 1][t - t0]
                if t == t0:
                    dec.decode_iframe_dev(coef)
                else:
                    dec.decode_pframe_dev(mv, has, coef)
            dctx.record(self.ev_dec[j & 1])

        def run(n_passes):
            units = [g for _ in range(n_passes) for g in gops]
            for j in range(len(units) + 1):
                if j < len(units):
                    if j >= 2:
                        ectx.wait_event(self.ev_dec[j & 1])      # the decode that last read this set of buffers is done
                    enc_gop(j, *units[j])
                if j >= 1:
                    dec_gop(j - 1, *units[j - 1])
        run(1)
        self.sync()
        t0 = time.perf_counter()
        run(reps)
        self.sync()
        el = time.perf_counter() - t0
        return reps * self.n_frames * self.S * self.n_mb / el

    def sync(self):
        self.ctx.sync()
        if self.dec_ctx is not None:
            self.dec_ctx.sync()

    def verify(self):
        self.sync()
        self.dec.check()
        assert np.array_equal(self.enc.prev_frame(), self.dec.framebuffer()), "decoder framebuffer != encoder reconstruction"

    def coded_fraction(self):
        h = np.empty(self.S * self.n_mb, np.uint8)
        self.ctx.download(h, self.has)
        return float(h.mean())

    def wall(self, reps, gop=GOP):
        """macroblocks/s of `reps` passes by the host clock (sync on both sides)"""
        self.step(gop)
        self.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.step(gop)
        self.sync()
        el = time.perf_counter() - t0
        return reps * self.n_frames * self.S * self.n_mb / el

    def close(self):
        self.sync()
        self.enc.close()
        self.dec.close()
        if self.ev_enc:
            for e in self.ev_enc:
                self.ctx.event_destroy(e)
            for e in self.ev_dec:
                self.dec_ctx.event_destroy(e)
        for p in self._bufs:
            self.ctx.free(p)
        self._bufs = []


class GopBatchSet:
    """ONE stream per GPU (or S of them), its frames resident in HBM in display order, and the GOPs of the stream -- not different
    videos -- in the slots of the sessions: encode_iframe never reads prev_frame and overwrites all of it (src/enc.rs:84-97),
    decode_plane_into overwrites the framebuffer (src/common.rs:477-496), so the I P P ... runs of a stream are independent and frame t of
    EVERY GOP goes into one launch per frame operation (include/pfv_hip.h: pfv_enc_session_set_frame_stride / _set_window).  The
    coefficients, motion vectors and frames produced are those of the serial frame-by-frame pass
    (tests: check_gop_batched_session; pfv_gop_encoder writes the same .pfv bytes); what changes is the launch shape: a 300-frame 4K stream is
    15 + 15 launches of 20 x 48 720 macroblocks instead of 300 + 300 launches of 48 720.  The decoded frames land in display order
    (strided fused crop)."""

    def __init__(self, pkg, ctx, W, H, Q, seeds, n_frames, gop=GOP, kind="pan", dec_ctx=None):
        """dec_ctx: a second context for the decoder (wall_pipelined): the decoder works through batch k - 1 while the encoder works through
        batch k, the two streams meet once per batch"""
        self.pkg, self.ctx, self.W, self.H, self.Q, self.S, self.n_frames, self.gop = pkg, ctx, W, H, Q, len(seeds), n_frames, gop
        self.seeds = [int(s) for s in seeds]
        self.kind, self.dec_ctx = kind, dec_ctx
        lib = pkg._lib.load()
        self.fb = fb = int(lib.pfv_frame_bytes(W, H))
        self.n_gops = (n_frames + gop - 1) // gop
        assert self.S == 1 or n_frames % gop == 0, "several streams: equal GOPs only (slot = stream x GOPs + GOP needs one stride)"
        self.slots = self.S * self.n_gops
        self.enc = pkg.EncoderSession(ctx, W, H, Q, self.slots)
        self.dec = pkg.DecoderSession(dec_ctx or ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), self.slots)
        self.n_mb = self.enc.total_blocks
        self.launch_streams = self.slots
        self._bufs = []
        n = self.slots
        self.ev_enc = self.ev_dec = None
        if dec_ctx is not None:      # two alternating sets of a batch's encode outputs (one buffer triple per frame step)
            one = lambda: (self._alloc(n * self.n_mb * 512), self._alloc(n * self.n_mb * 2), self._alloc(n * self.n_mb))
            self.pass_sets = [[one() for _ in range(min(n_frames, gop))] for _ in range(2)]
            self.ev_enc = [ctx.event(), ctx.event()]
            self.ev_dec = [dec_ctx.event(), dec_ctx.event()]
        self.frames = self._alloc(self.S * n_frames * fb)        # [stream][frame in display order][frame_bytes]
        self.out_frames = self._alloc(self.S * n_frames * fb)    # the decoded stream, display order
        self.coef, self.mv, self.has = self._alloc(n * self.n_mb * 512), self._alloc(n * self.n_mb * 2), self._alloc(n * self.n_mb)
        self.enc.set_frame_stride(gop * fb)
        for s, seed in enumerate(self.seeds):
            for t in range(n_frames):
                ctx.synth_frames_dev(W, H, [seed], t, self.frames + (s * n_frames + t) * fb, kind=kind)
        ctx.sync()

    def _alloc(self, n):
        p = self.ctx.alloc(max(int(n), 16))
        self._bufs.append(p)
        return p

    def active(self, t):
        """slots that have a frame t: all of them, or all but the last when the stream's last GOP is short"""
        if self.S > 1:
            return self.slots
        return sum(1 for g in range(self.n_gops) if g * self.gop + t < self.n_frames)

    def host_frames_at(self, stream, t):
        a = np.empty(self.fb, np.uint8)
        self.ctx.download(a, self.frames + (stream * self.n_frames + t) * self.fb)
        return a

    def host_frames(self, stream, count):
        return [self.host_frames_at(stream, t) for t in range(count)]

    def step(self, gop=None, on_launch=None, sample_frames=None):
        """the whole stream once: frame t of every GOP per launch, t = 0 .. gop - 1"""
        enc, dec, fb = self.enc, self.dec, self.fb
        for t in range(min(self.gop, self.n_frames)):
            n = self.active(t)
            if n != self.slots or t == 0:
                enc.set_window(0, n)
                dec.set_window(0, n)
            dec.set_output_strided_dev(self.out_frames + t * fb, self.gop * fb)
            f = self.frames + t * fb
            a = on_launch and on_launch()
            if t == 0:
                enc.encode_iframe_dev(f, self.coef)
                b = on_launch and on_launch()
                dec.decode_iframe_dev(self.coef)
                if on_launch:
                    on_launch("k_enc_iframe", a, b)
                    on_launch("k_dec_iframe", b, on_launch())
            else:
                enc.encode_pframe_dev(f, self.mv, self.has, self.coef)
                b = on_launch and on_launch()
                dec.decode_pframe_dev(self.mv, self.has, self.coef)
                if on_launch:
                    on_launch("k_enc_pframe", a, b)
                    on_launch("k_dec_pframe", b, on_launch())

    def encode_pack_pass(self, sets):
        enc, fb = self.enc, self.fb
        for t in range(min(self.gop, self.n_frames)):
            n = self.active(t)
            if n != self.slots or t == 0:
                enc.set_window(0, n)
            c, m, h = sets[t & 1]
            if t == 0:
                enc.encode_iframe_dev(self.frames, c)
                enc.pack_iframe_dev(c)
            else:
                enc.encode_pframe_dev(self.frames + t * fb, m, h, c)
                enc.pack_pframe_dev(m, h, c)

    def wall_pipelined(self, reps):
        """macroblocks/s of `reps` passes over the batch with the decoder ONE BATCH behind the encoder on its own context (host clock, both
        streams synchronised on both sides; every batch is encoded AND decoded inside the timed region)"""
        enc, dec, ectx, dctx, fb = self.enc, self.dec, self.ctx, self.dec_ctx, self.fb
        steps = min(self.gop, self.n_frames)

        def enc_batch(j):
            for t in range(steps):
                n = self.active(t)
                if n != self.slots or t == 0:
                    enc.set_window(0, n)
                coef, mv, has = self.pass_sets[j & 1][t]
                if t == 0:
                    enc.encode_iframe_dev(self.frames, coef)
                else:
                    enc.encode_pframe_dev(self.frames + t * fb, mv, has, coef)
            ectx.record(self.ev_enc[j & 1])

        def dec_batch(j):
            dctx.wait_event(self.ev_enc[j & 1])
            for t in range(steps):
                n = self.active(t)
                if n != self.slots or t == 0:
                    dec.set_window(0, n)
                dec.set_output_strided_dev(self.out_frames + t * fb, self.gop * fb)
                coef, mv, has = self.pass_sets[j & 1][t]
                if t == 0:
                    dec.decode_iframe_dev(coef)
                else:
                    dec.decode_pframe_dev(mv, has, coef)
            dctx.record(self.ev_dec[j & 1])

        def run(n):
            for j in range(n + 1):
                if j < n:
                    if j >= 2:
                        ectx.wait_event(self.ev_dec[j & 1])
                    enc_batch(j)
                if j >= 1:
                    dec_batch(j - 1)
        run(1)
        self.sync()
        t0 = time.perf_counter()
        run(reps)
        self.sync()
        return reps * self.n_frames * self.S * self.n_mb / (time.perf_counter() - t0)

    def sync(self):
        self.ctx.sync()
        if self.dec_ctx is not None:
            self.dec_ctx.sync()

    def verify(self):
        """decoder == encoder for every GOP that ran to the last step, and the display-order output holds each GOP's last frame"""
        self.sync()
        self.dec.check()
        n = self.active(min(self.gop, self.n_frames) - 1)
        a, b = self.enc.prev_frame(), self.dec.framebuffer()
        assert np.array_equal(a[:n], b[:n]), "decoder framebuffer != encoder reconstruction"
        last = np.empty(self.fb, np.uint8)
        t = min(self.gop, self.n_frames) - 1
        self.ctx.download(last, self.out_frames + t * self.fb)
        pf = self.pkg.VideoFrame.from_packed(self.W, self.H, b[0], padded=True)
        H, W = self.H, self.W
        want = np.concatenate([pf.plane_y.image()[:H, :W].reshape(-1), pf.plane_u.image()[:H // 2, :W // 2].reshape(-1),
                               pf.plane_v.image()[:H // 2, :W // 2].reshape(-1)])
        assert np.array_equal(last, want), "display-order output frame != crop of the framebuffer"

    def coded_fraction(self):
        h = np.empty(self.slots * self.n_mb, np.uint8)
        self.ctx.download(h, self.has)
        return float(h[: self.active(min(self.gop, self.n_frames) - 1) * self.n_mb].mean())

    def wall(self, reps, gop=None):
        self.step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.step()
        self.sync()
        el = time.perf_counter() - t0
        return reps * self.n_frames * self.S * self.n_mb / el

    def close(self):
        self.sync()
        self.enc.close()
        self.dec.close()
        if self.ev_enc:
            for e in self.ev_enc:
                self.ctx.event_destroy(e)
            for e in self.ev_dec:
                self.dec_ctx.event_destroy(e)
        for p in self._bufs:
            self.ctx.free(p)
        self._bufs = []


def entropy_side(ss, timer, args):
    """beside the headline (never part of `value`): the encoder alone with its entropy stage on the device, i.e. frames in
    HBM -> packet payloads in HBM (k_enc_* + k_ent_*), whole passes timed with HIP events"""
    enc, ctx, S, n_mb = ss.enc, ss.ctx, ss.S, ss.n_mb
    # two sets of encode outputs: with the stage on its own HIP stream the k_ent_* kernels of frame t overlap k_enc_pframe
    # of frame t+1; pack(t+1) orders later main-stream work behind pack(t)'s reads
    L = ss.launch_streams
    second = (ctx.alloc(L * n_mb * 512), ctx.alloc(L * n_mb * 2), ctx.alloc(L * n_mb))
    sets = [(ss.coef, ss.mv, ss.has), second]

    def encode_pass():
        ss.encode_pack_pass(sets)

    def measure(async_stream):
        enc.enable_entropy(async_stream=async_stream)
        encode_pass()
        enc.entropy_join()
        ctx.sync()
        reps = max(1, min(args.steps, 5))
        e0 = timer.stamp()
        for _ in range(reps):
            encode_pass()
        enc.entropy_join()          # the kernels' stream waits for the entropy stream before the closing event
        e1 = timer.stamp()
        ctx.sync()
        return timer.ms(e0, e1) / reps, enc.payload_sizes()

    serial_ms, sizes = measure(False)
    two_ms = None
    if not args.no_two_stream:
        two_ms, sizes_b = measure(True)
        assert np.array_equal(sizes, sizes_b), "entropy stage: two-stream and same-stream runs disagree"
        enc.enable_entropy(async_stream=False)
    for p in second:
        ctx.free(p)
    total = ss.n_frames * S * n_mb
    return {"value": total / (serial_ms * 1e-3), "unit": "macroblocks/s", "ms_per_pass": serial_ms,
            "two_stream_value": total / (two_ms * 1e-3) if two_ms else None, "two_stream_ms_per_pass": two_ms,
            "last_pframe_payload_bytes_per_stream": float(np.mean(sizes)),
            "note": "encode only, frames in HBM -> .pfv packet payloads in HBM: k_enc_iframe/k_enc_pframe + the device entropy "
                    "stage, HIP-event time over whole passes; two_stream_*: the stage on a second HIP stream with "
                    "double-buffered encode outputs"}


def low_motion_side(pkg, ctx, timer, W, H, Q, seeds, n_frames, default_kern_ms, default_coded, reps=4):
    """The same launch shape on LOW-MOTION content (static background, four moving noisy rectangles: about a quarter of a p-frame's
    macroblocks are coded, the rest skipped -- src/common.rs:221-222 transforms nothing for those): what the skip-aware transform of
    k_enc_pframe (tile-level compaction of the coded macroblocks) buys where there is something to skip."""
    L = pkg._lib
    off_ms = None
    if not EMU:     # the same content with the compaction switched off (sessions read the option when they are created)
        ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 0)
        try:
            ss0 = StreamSet(pkg, ctx, W, H, Q, seeds[:max(1, len(seeds) // 1)], n_frames, kind="low_motion")
            ss0.step()
            ctx.sync()
            marks = []
            ss0.step(on_launch=lambda name=None, a=None, b=None: timer.stamp() if name is None else (marks.append((a, b)) if name == "k_enc_pframe" else None))
            ctx.sync()
            off_ms = float(np.mean([timer.ms(a, b) for a, b in marks]))
            ss0.close()
        finally:
            ctx.set_option(L.PFV_OPT_TILE_COMPACTION, 1)
    floor_ms = None
    if not EMU:     # nothing coded at all (the background alone): what is left of k_enc_pframe is its search -- the floor of any skip-aware scheme
        ss0 = StreamSet(pkg, ctx, W, H, Q, seeds, min(n_frames, 3), kind="static")
        ss0.step()
        ctx.sync()
        marks = []
        ss0.step(on_launch=lambda name=None, a=None, b=None: timer.stamp() if name is None else (marks.append((a, b)) if name == "k_enc_pframe" else None))
        ctx.sync()
        floor_ms = float(np.mean([timer.ms(a, b) for a, b in marks]))
        floor_coded = ss0.coded_fraction()
        ss0.close()
    ss = StreamSet(pkg, ctx, W, H, Q, seeds, n_frames, kind="low_motion")
    ev = {k: [] for k in BYTES_PER_MB}

    def on_launch(name=None, a=None, b=None):
        if name is None:
            return timer.stamp()
        ev[name].append((a, b))
        return None
    ss.step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ss.step(on_launch=on_launch)
    ctx.sync()
    el = time.perf_counter() - t0
    ss.verify()
    coded = ss.coded_fraction()
    kern_ms = {k: float(np.mean([timer.ms(a, b) for a, b in v])) for k, v in ev.items() if v}
    launch_mbs = ss.S * ss.n_mb
    res = {"value": reps * n_frames * launch_mbs / el, "unit": "macroblocks/s (encode+decode)", "pframe_coded_fraction": round(coded, 4),
           "kernels_avg_launch_ms": kern_ms,
           "k_enc_pframe_vs_default_workload": kern_ms["k_enc_pframe"] / default_kern_ms["k_enc_pframe"] if "k_enc_pframe" in default_kern_ms else None,
           "k_enc_pframe_frac_of_hbm_peak": launch_mbs * BYTES_PER_MB_PENC / (kern_ms["k_enc_pframe"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "k_enc_pframe_ms_without_tile_compaction": off_ms,
           "k_enc_pframe_ms_nothing_coded": floor_ms, "nothing_coded_actual_coded_fraction": floor_coded if floor_ms else None,
           "k_enc_pframe_ms_perfect_compaction_bound": (floor_ms + coded * (default_kern_ms["k_enc_pframe"] - floor_ms) / default_coded) if floor_ms and default_coded else None,
           "note": "same streams-per-launch, geometry, quality and seeds as the headline; content kind low_motion (pfv_synth_frames_kind_dev). "
                   "nothing_coded: content kind static, every macroblock skipped = the search alone; perfect_compaction_bound = that floor + the "
                   "coded share of what transforming costs on the default workload (floor + coded * (default - floor) / default_coded): what a "
                   "transform that skipped every skipped macroblock at no cost would take"}
    ss.close()
    return res


def lookahead_side(pkg, ctx, Q):
    """What the throughput costs in BUFFERING (round-4 review, item 5).  The reference's API is frame at a time (src/enc.rs:75, :125,
    src/dec.rs:154-224); the device is only full when many frame operations share a launch.  Encode+decode macroblocks/s at kernel scope (host
    clock, frames resident) for ONE stream by how much of it is in flight: one frame (one launch per frame operation), one GOP (the same
    launches, decoder one GOP behind the encoder on a second context), 4 and 20 GOPs (frame t of every GOP per launch; `two_contexts`: the
    decoder one batch behind on a second context, priorities as in single_stream).  frames_in_flight = what a caller has to hold before
    the first packet / frame comes out."""
    out = {"unit": "macroblocks/s (encode+decode, kernel scope, one stream, quality %d)" % Q, "rows": []}
    ectx, dctx = pkg.Context(ctx.device, priority=1), pkg.Context(ctx.device, priority=-1)
    try:
        for name, W, H in (("1080p", 1920, 1080), ("4k", 3840, 2160)):
            seed = [pkg.synth.SEED]
            reps1 = 8 if name == "1080p" else 3
            ss = StreamSet(pkg, ctx, W, H, Q, seed, GOP)
            one = ss.wall(reps1)
            ss.close()
            ss = StreamSet(pkg, ectx, W, H, Q, seed, GOP, dec_ctx=dctx)
            one_gop = ss.wall_pipelined(4 * reps1)
            ss.verify()
            ss.close()
            out["rows"].append({"geometry": name, "in_flight": "1 frame", "frames_in_flight": 1, "value": one})
            out["rows"].append({"geometry": name, "in_flight": "1 GOP (decoder one GOP behind, two contexts)", "frames_in_flight": 2 * GOP, "value": one_gop})
            for k in (4, 20):
                reps = max(2, (24 if name == "1080p" else 6) // k * 2)
                gb = GopBatchSet(pkg, ctx, W, H, Q, seed, k * GOP)
                plain = gb.wall(reps)
                gb.verify()
                gb.close()
                row = {"geometry": name, "in_flight": f"{k} GOPs per launch", "frames_in_flight": k * GOP, "value": plain}
                if k == 4:      # two sets of 15 x k slots of encode outputs: 1.5 GB (4K); not worth 15 GB for 20 GOPs, whose launches fill the device already
                    gb = GopBatchSet(pkg, ectx, W, H, Q, seed, k * GOP, dec_ctx=dctx)
                    row["two_contexts"] = gb.wall_pipelined(2 * reps)
                    row["two_contexts_frames_in_flight"] = 2 * k * GOP
                    gb.verify()
                    gb.close()
                out["rows"].append(row)
    finally:
        dctx.close()
        ectx.close()
    return out


def single_stream_side(pkg, ctx, Q, reps=6):
    """The reference's caller is ONE Encoder per stream (src/enc.rs:125-173): what a single 1080p stream (and 8 of them)
    gets at kernel scope, with one launch per frame operation, with a whole GOP replayed as one HIP graph, and with the
    decoder on its own context (second HIP stream) one GOP behind the encoder: GOP g is decoded while GOP g + 1 is encoded,
    the streams meet once per GOP (a per-FRAME hand-over between the streams was measured too: slower than one stream, the
    cross-queue signalling costs more than a 1080p frame's kernels take)."""
    out = {}
    ectx, dctx = pkg.Context(ctx.device, priority=1), pkg.Context(ctx.device, priority=-1)     # two-context schedule: encoder stream high, decoder stream low
    for S in (1, 8):
        seeds = [pkg.synth.SEED + 17 * k for k in range(S)]
        ss = StreamSet(pkg, ctx, 1920, 1080, Q, seeds, GOP)
        r = {"launches": ss.wall(reps)}
        ss.verify()
        ss2 = StreamSet(pkg, ectx, 1920, 1080, Q, seeds, GOP, dec_ctx=dctx)
        r["decoder_one_gop_behind_on_second_stream"] = ss2.wall_pipelined(4 * reps)
        ss2.verify()
        ss2.close()
        try:
            r["hip_graph"] = graph_rate(ss, reps)
            ss.verify()
        except Exception as e:      # noqa: BLE001 -- side measurement: report, do not fail the bench
            r["hip_graph_error"] = str(e)[:200]
        out[f"streams_{S}"] = r
        ss.close()
    dctx.close()
    ectx.close()
    # one 1080p stream of 300 frames with its 20 GOPs as the slots of each launch (GopBatchSet; round 4): what the reference's own
    # usage -- one Encoder, one stream -- gets when the caller has more than one GOP in hand
    gb = GopBatchSet(pkg, ctx, 1920, 1080, Q, [pkg.synth.SEED], 20 * GOP)
    out["streams_1"]["gop_batched_20_gops_per_launch"] = gb.wall(reps)
    gb.verify()
    gb.close()
    out["unit"] = "macroblocks/s (encode+decode, 1080p GOP-15, kernel scope, host clock incl. launch overhead)"
    return out


def graph_rate(ss, reps):
    """one GOP (15 x encode + decode launches, first frame an i-frame) recorded once, replayed as one graph launch per GOP"""
    with ss.pkg.Graph(ss.ctx) as g:
        ss.step()
    g.launch()
    ss.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.launch()
    ss.ctx.sync()
    el = time.perf_counter() - t0
    g.close()
    return reps * ss.n_frames * ss.S * ss.n_mb / el


def stream_4k_side(pkg, ctx, Q, seed, n_frames=300, pcie_frames=30, ss=None, gop_batched_rate=None, W=3840, H=2160):
    """BASELINE config #4 at the three scopes of SURVEY.md section 8d: one 3840x2160 GOP-15 stream; (i) kernels only, frames
    and coefficients resident in HBM -- one launch per frame operation, and GOP-batched (frame t of every GOP per launch); (ii) + PCIe
    through the host-buffer session entry points on page-locked buffers (a 30-frame sample); (iii) end to end, ALL frames: GopEncoder
    -> .pfv bytes -> GopDecoder (producer frames in page-locked memory, uploads on a copy stream under the previous batch's kernels,
    device entropy stage; GOP-parallel host bit parser on the way back), with the frame-by-frame Encoder / Decoder objects beside them
    (same bytes).  The producer's frames (generated on the device, brought to page-locked host memory) are outside the timed calls; a
    decoded frame is checked against the closed-loop reconstruction of the session path."""
    own = ss is None
    if own:
        ss = StreamSet(pkg, ctx, W, H, Q, [seed], n_frames)
    n_frames = ss.n_frames
    res = {"config": f"{W}x{H}, {n_frames} frames, GOP-{GOP}, quality {Q}, seed {seed}", "macroblocks_per_frame": ss.n_mb}
    res["kernel_only"] = {"value": ss.wall(2), "frames": n_frames,
                          "note": "one launch per frame operation, 48 720 macroblocks per launch (the serial pass)"}
    ss.verify()
    if gop_batched_rate is None and own:
        ss.close()                                        # one resident copy of the 3.7 GB of frames at a time
        gb = GopBatchSet(pkg, ctx, W, H, Q, [seed], n_frames)
        gop_batched_rate = gb.wall(2)
        gb.verify()
        gb.close()
        ss = StreamSet(pkg, ctx, W, H, Q, [seed], n_frames)
    res["kernel_only"]["gop_batched"] = {"value": gop_batched_rate, "gops_per_launch": (n_frames + GOP - 1) // GOP,
                                         "note": "frame t of every GOP of the stream in one launch per frame operation (GopBatchSet): same coefficients and "
                                                 "frames, 15 + 15 launches of 20 x 48 720 macroblocks instead of 300 + 300 of 48 720"}
    if own:     # the same stream with the decoder on its own context (second HIP stream), one GOP behind the encoder
        ectx, dctx = pkg.Context(ctx.device, priority=1), pkg.Context(ctx.device, priority=-1)
        ss2 = StreamSet(pkg, ectx, W, H, Q, [seed], min(n_frames, 60), dec_ctx=dctx)
        res["kernel_only"]["decoder_one_gop_behind_on_second_stream"] = ss2.wall_pipelined(4)
        ss2.verify()
        ss2.close()
        dctx.close()
        ectx.close()
    # (ii) + PCIe: the host-buffer session entry points on PAGE-LOCKED buffers (pfv_host_alloc), synchronous per frame
    import ctypes
    lib = pkg._lib.load()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    pcie_frames = min(pcie_frames, n_frames)
    fbytes, n_mb = ss.fb, ss.n_mb
    host_all = ctx.host_array(n_frames * fbytes).reshape(n_frames, fbytes)      # the producer's frames, page-locked, display order
    ctx.download(host_all.reshape(-1), ss.frame_ptr(0) if hasattr(ss, "frame_ptr") else ss.frames)
    enc = pkg.EncoderSession(ctx, W, H, Q, 1)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), 1)
    pin = {k: ctx.host_array(n) for k, n in (("coef", n_mb * 512), ("mv", n_mb * 2), ("has", n_mb), ("out", fbytes))}
    qi, qp = np.array([0, 1, 1], np.uint8), np.array([2, 3, 3], np.uint8)
    t0 = time.perf_counter()
    for t in range(pcie_frames):
        if t % GOP == 0:
            ctx.check(lib.pfv_enc_iframe(enc.handle, P(host_all[t]), P(pin["coef"])))
            ctx.check(lib.pfv_dec_iframe(dec.handle, P(pin["coef"]), P(qi)))
        else:
            ctx.check(lib.pfv_enc_pframe(enc.handle, P(host_all[t]), P(pin["mv"]), P(pin["has"]), P(pin["coef"])))
            ctx.check(lib.pfv_dec_pframe(dec.handle, P(pin["mv"]), P(pin["has"]), P(pin["coef"]), P(qp)))
        ctx.check(lib.pfv_dec_get_frame(dec.handle, P(pin["out"])))
    el = time.perf_counter() - t0
    res["pcie_inclusive"] = {"value": pcie_frames * n_mb / el, "frames": pcie_frames,
                             "note": "host-buffer session entry points on page-locked buffers (pfv_host_alloc): frame up, coefficients down and up "
                                     "again, decoded frame down; synchronous per frame"}
    recon_last = enc.prev_frame()[0]
    enc.close()
    dec.close()
    for a in pin.values():
        ctx.host_free(a)
    if own:
        ss.close()
    pf = pkg.VideoFrame.from_packed(W, H, recon_last, padded=True)     # frame pcie_frames - 1 as the session path reconstructs it, cropped
    want = np.concatenate([pf.plane_y.image()[:H, :W].reshape(-1), pf.plane_u.image()[:H // 2, :W // 2].reshape(-1),
                           pf.plane_v.image()[:H // 2, :W // 2].reshape(-1)])
    ny, nc = W * H, (W // 2) * (H // 2)
    facts = host_cpu_facts()
    quota = facts["cgroup_cpu_quota"] if isinstance(facts["cgroup_cpu_quota"], (int, float)) else (facts["affinity_cpus"] or facts["host_cpus"])
    parse_threads = int(max(1, min(32, round(quota)) - 1))

    # (iii) end to end, ALL frames, through the GOP-batched objects: page-locked planes -> copy stream -> k_enc_* + device entropy stage
    # for frame t of every GOP of a batch -> .pfv bytes; .pfv bytes -> GOP-parallel packet parse -> k_dec_* -> frames in page-locked memory
    class Chunks:                                          # a writer that keeps what it is handed (joined outside the timed region)
        def __init__(self):
            self.parts, self.n = [], 0

        def write(self, b):
            self.parts.append(bytes(b))
            self.n += len(b)

    class Count:                                           # a writer that consumes (a socket, a file): the bytes are checked in the run beside it
        def __init__(self):
            self.n = 0

        def write(self, b):
            self.n += len(b)

    def run_encoder(make_enc, raw, sink):
        e = make_enc(sink)
        t0 = time.perf_counter()
        for t in range(n_frames):
            f = host_all[t]
            if raw:
                (e.encode_iframe if t % GOP == 0 else e.encode_pframe)((f[:ny], f[ny:ny + nc], f[ny + nc:]))
            else:
                (e.encode_iframe if t % GOP == 0 else e.encode_pframe)(pkg.VideoFrame.from_packed(W, H, f))
        e.finish()
        el = time.perf_counter() - t0
        st = e.stats() if hasattr(e, "stats") else None
        e.close()
        return el, st

    def run_decoder(make_dec, data, raw):
        d = make_dec(data)
        n, ok = [0], [None]

        def onvideo(*fr):
            n[0] += 1
            if n[0] == pcie_frames:
                got = np.concatenate(fr) if raw else fr[0].packed()
                ok[0] = bool(np.array_equal(got, want))
        t0 = time.perf_counter()
        while d.advance_frame(onvideo):
            pass
        el = time.perf_counter() - t0
        st = d.stats() if hasattr(d, "stats") else None
        d.close()
        assert n[0] == n_frames and ok[0], "decoded .pfv frame != encoder reconstruction"
        return el, st

    gops = 10
    mk_gop = lambda sink, zc: pkg.GopEncoder(sink, W, H, 30, Q, ctx, max_gops=gops, max_gop_frames=GOP, zero_copy=zc)
    counted = Count()
    t_enc, st_e = run_encoder(lambda sink: mk_gop(sink, True), True, counted)       # the timed pass: the writer consumes the segments in place
    kept = Chunks()
    run_encoder(lambda sink: mk_gop(sink, False), True, kept)                        # the same again with a writer that keeps the bytes, to check them
    data = b"".join(kept.parts)
    assert counted.n == len(data)
    mk_dec = lambda mode: (lambda data: pkg.GopDecoder(data, ctx, max_gops=2 * gops, max_gop_frames=GOP, threads=parse_threads, raw=True, entropy=mode))
    t_dec_host, st_d_host = run_decoder(mk_dec("host"), data, True)
    run_decoder(mk_dec("device"), data, True)                                       # first use: allocations, code objects
    t_dec, st_d = run_decoder(mk_dec("device"), data, True)
    res["end_to_end"] = {"frames": n_frames, "stream_bytes": len(data), "bits_per_pixel": round(len(data) * 8 / (n_frames * W * H), 3),
                         "encode_value": n_frames * n_mb / t_enc, "decode_value": n_frames * n_mb / t_dec,
                         "value": n_frames * n_mb / (t_enc + t_dec), "gops_per_batch": {"encoder": gops, "decoder": 2 * gops}, "parse_threads": parse_threads,
                         "upload_GBps_equivalent": n_frames * fbytes / t_enc / 1e9, "encode_s": t_enc, "decode_s": t_dec,
                         "encoder_host_seconds": st_e, "decoder_host_seconds": st_d,
                         "decode_payloads_read_on_host": {"decode_value": n_frames * n_mb / t_dec_host, "decode_s": t_dec_host, "decoder_host_seconds": st_d_host,
                                                          "note": "PFV_ENTROPY_DECODE_HOST: the packets of a step parsed by the host pool (round 4's first form)"},
                         "note": "GopEncoder -> .pfv bytes -> GopDecoder (pfv_gop_encoder / pfv_gop_decoder: frame t of every GOP of a batch per launch), every "
                                 "frame of the stream; producer frames in page-locked memory (uploaded on a copy stream under the previous batch's kernels), "
                                 "packets handed to the writer as segments where they lie (pfv_gop_encoder_drain_iov); packets parsed GOP-parallel, decoded "
                                 "frames delivered from page-locked memory; stream bytes == the serial Encoder's, a decoded frame checked against the "
                                 "encoder's reconstruction"}
    skept = Chunks()
    s_enc, _ = run_encoder(lambda sink: pkg.Encoder(sink, W, H, 30, Q, ctx), False, skept)
    sdata = b"".join(skept.parts)
    assert sdata == data, "GOP-batched and serial encoder objects wrote different .pfv streams"
    s_dec, _ = run_decoder(lambda data: pkg.Decoder(data, ctx), sdata, False)
    res["end_to_end"]["serial_objects"] = {"encode_value": n_frames * n_mb / s_enc, "decode_value": n_frames * n_mb / s_dec,
                                           "note": "Encoder -> .pfv -> Decoder, one frame per call and per launch (the same bytes: checked)"}
    ctx.host_free(host_all.reshape(-1))
    res["end_to_end"]["native_host"] = native_end_to_end(W, H, n_frames, Q, gops, parse_threads)      # its own context; this one's page-locked frames are gone
    res["unit"] = "macroblocks/s"
    return res


def native_end_to_end(W, H, n_frames, Q, gops, parse_threads):
    """the same end-to-end run from a native host program over the C ABI (tools/e2e_native.cpp, built here with g++): what a compiled
    caller -- the reference's are Rust -- sees, without the Python mirror's ~0.15 ms of interpreter per delivered frame"""
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.gettempdir(), f"pfv_e2e_native_{os.getpid()}")
    libdir = os.path.join(ROOT, "pretty-fast-video_amd")
    try:
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "e2e_native.cpp"), "-L", libdir, "-lpfv_hip",
                        f"-Wl,-rpath,{libdir}", "-o", exe], check=True, capture_output=True, text=True)
        r = subprocess.run([exe, str(W), str(H), str(n_frames), str(GOP), str(Q), str(gops), str(2 * gops), str(parse_threads)], check=True, capture_output=True,
                           text=True, timeout=600)
        out = json.loads(r.stdout)
        out["note"] = ("tools/e2e_native.cpp: page-locked producer frames -> pfv_gop_encoder -> .pfv bytes -> pfv_gop_decoder -> the consumer's callback per frame "
                       "(best of 3 per decoder mode; sampled frames identical in all modes)")
        # config 4's clip is ONE batch of the encoder (its first and its last): the same call sequence on three times the stream, where the
        # start of the object and the last download of a batch are off the critical path (profiles/r06_byref_floor.md, section 4)
        try:
            r = subprocess.run([exe, str(W), str(H), str(3 * n_frames), str(GOP), str(Q), str(gops), str(2 * gops), str(parse_threads)], check=True,
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, PFV_E2E_STOP_AFTER_ENCODE="1"))
            d = json.loads(r.stdout)
            out["three_times_the_stream"] = {"frames": 3 * n_frames, "stream_bytes": d["stream_bytes"], "encode_value_frames_in_hbm": d["encode_value_frames_in_hbm"],
                                             "encode_value_frames_in_hbm_by_reference": d["by_reference"], "encode_frames_in_hbm_by_reference_s": d["by_reference_s"],
                                             "note": "the encoder objects alone (no decode), same batch width: batches behind the first run at the kernels' rate"}
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
            out["three_times_the_stream"] = {"error": f"{type(e).__name__}: {getattr(e, 'stderr', '') or e}"[:300]}
        return out
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": f"{type(e).__name__}: {getattr(e, 'stderr', '') or e}"[:400]}
    finally:
        if os.path.exists(exe):
            os.remove(exe)


def batch_encoder_side(pkg, ctx, Q, n_streams=32, reps=3):
    """End to end beyond north_star's boundary: n 1080p streams through the C++ batch encoder -- producer frames in
    page-locked host memory -> PCIe -> k_enc_* + device entropy stage -> payloads -> PCIe -> .pfv packets at in-memory writers.
    The upload of step t+1 runs on a copy stream under the kernels and the host-side collection of step t."""
    W, H = 1920, 1080
    fb = int(pkg._lib.load().pfv_frame_bytes(W, H))
    seeds = [pkg.synth.SEED + 17 * k for k in range(n_streams)]
    dev = ctx.alloc(n_streams * fb)
    host = []
    for t in range(GOP):                                   # the producer's frames, already in page-locked memory
        a = ctx.host_array(n_streams * fb)
        ctx.synth_frames_dev(W, H, seeds, t, dev)
        ctx.download(a, dev)
        host.append(a.reshape(n_streams, fb))
    ctx.free(dev)

    class Sink:                                            # writer that counts (the bytes are checked in tests/, not here)
        def __init__(self):
            self.n = 0

        def write(self, b):
            self.n += len(b)
    sinks = [Sink() for _ in range(n_streams)]
    be = pkg.BatchEncoder(sinks, W, H, 30, Q, ctx)
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        for t in range(GOP):
            (be.encode_iframes if t == 0 else be.encode_pframes)(host[t])
        be.flush()
        best = max(best, GOP * n_streams * 12240 / (time.perf_counter() - t0))
    be.close()
    res = {"value": best, "unit": "macroblocks/s", "streams": n_streams, "stream_bytes_per_gop": sinks[0].n // reps,
           "upload_GBps_equivalent": best / 12240 * fb / 1e9,
           "note": "1080p GOP-15, encode only, best of %d passes; never part of `value`" % reps}
    # the way back: the same n streams (2 GOPs each, kept in memory this time) through the C++ batch decoder
    bufs = [io.BytesIO() for _ in range(n_streams)]
    be = pkg.BatchEncoder(bufs, W, H, 30, Q, ctx)
    for t in range(2 * GOP):
        (be.encode_iframes if t % GOP == 0 else be.encode_pframes)(host[t % GOP])
    be.close()
    data = [b.getvalue() for b in bufs]
    dec, dec_dev = {}, {}
    for mode, ths, out in (("host", (8, 16, 32), dec), ("device", (4, 8), dec_dev)):
        for th in ths:
            bd = pkg.BatchDecoder(data, ctx, threads=th, entropy=mode)
            t0 = time.perf_counter()
            steps = 0
            while bd.advance_frames() is not False:
                steps += 1
            el = time.perf_counter() - t0
            assert steps == 2 * GOP and bd.dense_steps == 0
            counts = bd.entropy_counts()
            assert mode == "host" or counts["packets_read_on_device"] == steps * n_streams, counts
            bd.close()
            out[str(th)] = steps * n_streams * 12240 / el
    res["batch_decoder"] = {"value": max(dec_dev.values()), "unit": "macroblocks/s", "by_threads": dec_dev,
                            "payloads_read_on_host": {"value": max(dec.values()), "by_parse_threads": dec},
                            "note": ".pfv bytes in host memory -> tables and block headers read on a worker pool (step t+1 under the device work "
                                    "of step t), run streams read by the device's entropy stage (k_entd_*) -> k_dec_* -> frames back in "
                                    "page-locked host memory; payloads_read_on_host: the host bit parser on the pool -> coefficient lists "
                                    "read by the scatter kernel (round 3's form)"}
    return res


def config23_side(pkg, ctx, timer, Q, with_cpu=True, reps=200):
    """BASELINE.md section 3's rows for configs #2 and #3 as written, in the reference's own shape (src/lib.rs:241-252, test_encode_1: ONE
    image, encode_iframe then encode_pframe):
      config2 -- ONE 1080p i-frame (frame 0 of the synthetic pan stream): 8x8 DCT + quantise (k_enc_iframe, with the closed-loop
                 reconstruction of src/enc.rs:84-97) and dequantise + iDCT (k_dec_iframe, with the fused retframe crop), one launch of 12 240
                 macroblocks each;
      config3 -- ONE 1080p p-frame encode (motion search + residual DCT + reconstruction, k_enc_pframe) of frame 1 against the
                 reconstruction of frame 0: the translated-noise pair of SURVEY 8d (the texture translated by (3, 2) luma pixels between the
                 two frames, +-16 noise on about half of the macroblocks).
    Per kernel: mean and minimum HIP-event time of `reps` single launches (events on the context's stream; a 12 240-macroblock launch is
    10-20 microseconds, of which ~4 are dispatch + completion -- that is what ONE frame costs, the batched figures are in `kernels`),
    macroblocks/s, algorithmic GB/s (SURVEY 8d bytes; + 256 B for the crop the decoder fuses) and the fraction of the 8 TB/s peak; the CPU
    oracle on the same frames at 1 thread and at usable_cpus threads, and GPU / CPU."""
    W, H = 1920, 1080
    ss = StreamSet(pkg, ctx, W, H, Q, [pkg.synth.SEED], 2)
    enc, dec, n_mb = ss.enc, ss.dec, ss.n_mb
    f0, f1 = ss.frame_ptr(0), ss.frame_ptr(1)
    ev = {"k_enc_iframe": [], "k_dec_iframe": [], "k_enc_pframe": []}
    timer.reserve(4 * reps + 8)
    for r in range(reps + 3):
        a = timer.stamp()
        enc.encode_iframe_dev(f0, ss.coef)
        b = timer.stamp()
        dec.decode_iframe_dev(ss.coef)
        c = timer.stamp()
        enc.encode_pframe_dev(f1, ss.mv, ss.has, ss.coef)
        d = timer.stamp()
        if r >= 3:          # the first launches pay for code-object loading
            ev["k_enc_iframe"].append((a, b)); ev["k_dec_iframe"].append((b, c)); ev["k_enc_pframe"].append((c, d))
    ctx.sync()
    coded = ss.coded_fraction()
    us = {k: np.array([timer.ms(a, b) for a, b in v]) * 1e3 for k, v in ev.items()}
    # the same launches back to back (no event between them): what the launch costs when the queue stays full
    def train(fn, n=reps):
        fn(); ctx.sync()
        a = timer.stamp()
        for _ in range(n):
            fn()
        b = timer.stamp()
        ctx.sync()
        return timer.ms(a, b) * 1e3 / n
    btb = {"k_enc_iframe": train(lambda: enc.encode_iframe_dev(f0, ss.coef)), "k_dec_iframe": train(lambda: dec.decode_iframe_dev(ss.coef))}
    # p-frame launches: a second p-frame of the same image would find its own reconstruction and skip everything, so the train is (i-frame,
    # p-frame) pairs and the i-frame train is taken off
    def pair():
        enc.encode_iframe_dev(f0, ss.coef)
        enc.encode_pframe_dev(f1, ss.mv, ss.has, ss.coef)
    btb["k_enc_pframe"] = train(pair) - btb["k_enc_iframe"]
    host = ss.host_frames(0, 2) if with_cpu else None
    ss.close()

    def row(k):
        t_us = float(us[k].mean())
        return {"kernel": k, "us_per_launch_mean": t_us, "us_per_launch_min": float(us[k].min()), "us_per_launch_back_to_back": btb[k],
                "macroblocks_per_s": n_mb / (t_us * 1e-6), "algorithmic_bytes_per_macroblock": BYTES_PER_MB[k],
                "algorithmic_GBps": n_mb * BYTES_PER_MB[k] / (t_us * 1e-6) / 1e9, "pct_of_hbm_peak": 100.0 * n_mb * BYTES_PER_MB[k] / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_GBps_survey_bytes": n_mb * BYTES_PER_MB_SURVEY[k] / (t_us * 1e-6) / 1e9}
    c2 = {"config": "BASELINE config #2: ONE 1920x1080 i-frame, quality %d: DCT + quantise (+ closed-loop reconstruction) and dequantise + iDCT" % Q,
          "macroblocks": n_mb, "launches_timed": reps, "encode": row("k_enc_iframe"), "decode": row("k_dec_iframe")}
    rt_us = c2["encode"]["us_per_launch_mean"] + c2["decode"]["us_per_launch_mean"]
    c2["round_trip"] = {"us": rt_us, "macroblocks_per_s": n_mb / (rt_us * 1e-6),
                        "algorithmic_GBps": n_mb * (BYTES_PER_MB_SURVEY["k_enc_iframe"] + BYTES_PER_MB_SURVEY["k_dec_iframe"]) / (rt_us * 1e-6) / 1e9}
    c2["round_trip"]["pct_of_hbm_peak"] = 100.0 * c2["round_trip"]["algorithmic_GBps"] / HBM_PEAK_GBS
    c3 = {"config": "BASELINE config #3: ONE 1920x1080 p-frame encode (motion search + residual DCT + reconstruction), quality %d, synthetic two-frame "
                    "translated-noise pair (frames 0 and 1 of the pan stream)" % Q,
          "macroblocks": n_mb, "launches_timed": reps, "pframe_coded_fraction": round(coded, 4), "encode": row("k_enc_pframe")}
    note = ("single launches of ONE frame (12 240 macroblocks = 1.5 wavefronts per SIMD): launch latency, not throughput -- the library picks the "
            "16-lanes-per-macroblock mapping at this grid size; the batched launch shape (96 frames per launch) is in `kernels` / `roofline`")
    c2["note"] = c3["note"] = note
    if with_cpu:
        from oracle_bind import Oracle, OracleDecoder
        ora = Oracle()
        tabs = np.stack(ora.qtables(Q)[:4])
        usable = usable_cpus(host_cpu_facts())

        def cpu(threads, n=5):
            ora.L.pfvo_pool_shutdown()
            e = ora.encoder(W, H, Q, threads=threads)
            d = OracleDecoder(ora, W, H, tabs, threads=threads)
            ti, td, tp = [], [], []
            for _ in range(n):
                t0 = time.perf_counter(); c = e.encode_iframe(host[0])
                t1 = time.perf_counter(); d.decode_iframe(c)
                t2 = time.perf_counter(); e.encode_pframe(host[1])
                t3 = time.perf_counter()
                ti.append(t1 - t0); td.append(t2 - t1); tp.append(t3 - t2)
            med = lambda v: float(np.median(v))
            return {"threads": threads, "iframe_encode_ms": med(ti) * 1e3, "iframe_decode_ms": med(td) * 1e3, "pframe_encode_ms": med(tp) * 1e3,
                    "iframe_round_trip_macroblocks_per_s": n_mb / (med(ti) + med(td)), "pframe_encode_macroblocks_per_s": n_mb / med(tp)}
        one, allt = cpu(1), cpu(usable)
        ora.L.pfvo_pool_shutdown()
        c2["cpu_oracle"] = {"at_1_thread": {k: one[k] for k in ("threads", "iframe_encode_ms", "iframe_decode_ms", "iframe_round_trip_macroblocks_per_s")},
                            "at_usable_cpus": {k: allt[k] for k in ("threads", "iframe_encode_ms", "iframe_decode_ms", "iframe_round_trip_macroblocks_per_s")},
                            "kind": "port (median of 5 frames; the reference's oracle restatement, not the Rust binary)"}
        c2["gpu_over_cpu"] = {"vs_1_thread": c2["round_trip"]["macroblocks_per_s"] / one["iframe_round_trip_macroblocks_per_s"],
                              "vs_usable_cpus": c2["round_trip"]["macroblocks_per_s"] / allt["iframe_round_trip_macroblocks_per_s"]}
        c3["cpu_oracle"] = {"at_1_thread": {k: one[k] for k in ("threads", "pframe_encode_ms", "pframe_encode_macroblocks_per_s")},
                            "at_usable_cpus": {k: allt[k] for k in ("threads", "pframe_encode_ms", "pframe_encode_macroblocks_per_s")},
                            "kind": "port (median of 5 frames)"}
        c3["gpu_over_cpu"] = {"vs_1_thread": c3["encode"]["macroblocks_per_s"] / one["pframe_encode_macroblocks_per_s"],
                              "vs_usable_cpus": c3["encode"]["macroblocks_per_s"] / allt["pframe_encode_macroblocks_per_s"],
                              "north_star_target": ">= 50 x the reference CPU macroblocks/s on 1080p p-frame encode (a ratio, never credit for kernel quality: see roofline)"}
    return c2, c3


def traffic_from_profiles(S, W, H, Q):
    """HBM traffic per k_enc_pframe launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
    tools/gpu_pmc.sh).  PMC counters cannot be collected inside this process; the value is quoted only when it was
    collected on this exact configuration, and the line says where it came from."""
    try:
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pm = json.load(open(path))
        c = pm["config"]
        if (int(c["streams"]), int(c["width"]), int(c["height"]), int(c["quality"])) != (S, W, H, Q):
            return None, None, None, None
        k = pm["kernels"]["k_enc_pframe"]
        import __graft_entry__ as graft
        built, here = pm.get("build_id"), graft.hip_build_id()
        src = f"profiles/pmc_traffic.json (rocprofv3 --pmc passes on build {built}; this run's library is build {here}" + \
              (")" if built == here else " -- a different build of the kernels)")
        return k["traffic_bytes"], k.get("valu_wave_instructions"), src + "; quoted, not measured in this run", k.get("wavefronts")
    except (OSError, KeyError, ValueError):
        return None, None, None, None


def live_pmc(args, S, W, H, Q, NF, per_pass_timeout=60):
    """HBM traffic and VALU instruction count of k_enc_pframe on THIS workload, measured now: three short child runs of this script under
    `rocprofv3 --kernel-trace --pmc <counter>` (one counter group per run, as MI355X_MICROARCH.md prescribes: FETCH_SIZE, WRITE_SIZE,
    SQ_INSTS_VALU + SQ_WAVES), parsed like tools/pmc_traffic.py (FETCH_SIZE in KiB and doubled: the gfx950 half-count correction).
    None when rocprofv3 is missing or a pass fails / times out -- the line then quotes profiles/pmc_traffic.json as before."""
    import csv
    import re
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    # already running under a profiler (rocprofv3 -- python bench.py): do not nest one inside it
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-two-stream", "--no-extra", "--no-entropy",
             "--no-live-pmc", "--no-verify", "--streams", str(S), "--width", str(W), "--height", str(H), "--frames", str(NF), "--quality", str(Q),
             "--workload", args.workload] + (["--serial-gops"] if args.serial_gops else [])
    out = {}
    tmp = tempfile.mkdtemp(prefix="pfv_pmc_")
    try:
        for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_INSTS_VALU", "SQ_WAVES"])):
            d = os.path.join(tmp, name)
            env = dict(os.environ, TMPDIR=tmp)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", *counters, "-f", "csv", "-d", d, "-o", name, "--", *child], cwd=tmp, env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=per_pass_timeout)
            if r.returncode != 0:
                return None
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if not files:
                return None
            acc = {}
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    m = re.search(r"pfv::(k_enc_pframe|k_enc_iframe|k_dec_pframe|k_dec_iframe)\b", row.get("Kernel_Name", ""))
                    if m:
                        acc.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c in counters:
                if not acc.get("k_enc_pframe", {}).get(c):
                    return None
                for kname, by_counter in acc.items():
                    if by_counter.get(c):
                        out.setdefault(kname, {})[c] = sum(by_counter[c]) / len(by_counter[c])
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    def one(o):
        return {"traffic_bytes": o["FETCH_SIZE"] * 1024 * 2 + o["WRITE_SIZE"] * 1024, "read_bytes": o["FETCH_SIZE"] * 1024 * 2,
                "write_bytes": o["WRITE_SIZE"] * 1024, "valu_wave_instructions": o["SQ_INSTS_VALU"], "wavefronts": o["SQ_WAVES"]}
    res = one(out["k_enc_pframe"])
    res["kernels"] = {k: one(o) for k, o in out.items() if all(c in o for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES"))}
    return res


# ---------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    t_start = time.perf_counter()
    sections = {}                                     # wall seconds of each part of this run (rank 0), for auditing the run's budget

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL between processes needs dmabuf IPC on this driver stack (hipIpcGetMemHandle fails otherwise); the launcher normally exports it, a
    # rank started some other way sets it before the HIP runtime is loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import __graft_entry__ as graft
    # No torch anywhere: its wheel bundles a second HIP / HSA runtime, and with both runtimes in one process every launch of a
    # small kernel costs about twice as much host time (measured in round 2: a 4K single-stream pass of 600 launches takes 14.2 ms
    # without `import torch`, 25.7 ms with it).  The ranks of an N > 1 job talk through pretty-fast-video_amd/comm.py: RCCL via the
    # library on the context's own stream, the ncclUniqueId over a TCP rendezvous.
    stdout_fd = None
    if world > 1 or args.force_comm:
        # the contract is ONE line on stdout: anything a library prints on fd 1 (RCCL's version banner) goes to stderr
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)

    pkg = graft.load_package()
    if EMU:
        import conftest                               # tests/conftest.py: g++ build of the kernel sources on the fiber emulator
        import libswitch
        libswitch.use(pkg, conftest.build_emulator())
        share, local_rank = True, 0
    else:
        graft.build_hip()
        share = False
        if os.environ.get("PFV_HIP_LIB"):             # A/B scripts under tools/: a variant build of the kernels (the line says so: `library`)
            import libswitch
            libswitch.apply_from_env(pkg)
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    commlib = import_module("pretty_fast_video_amd.comm")
    if world > 1 and not EMU:
        # rank -> GPU: the LOCAL_RANK-th VISIBLE device (HIP numbers the devices HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES leave, in that
        # order), one rank per device.  With ONE visible device (developer dry run of the N > 1 control flow on a one-GPU box, or
        # PFV_BENCH_SHARE_GPU=1) every rank uses it and the control plane stays on the rendezvous sockets: RCCL refuses two ranks on a
        # device.  Anything in between is a misconfigured launch and says so instead of piling ranks onto some of the GPUs.
        n_dev = int(pkg._lib.load().pfv_device_count())
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        share = os.environ.get("PFV_BENCH_SHARE_GPU") == "1" or n_dev == 1
        if share:
            local_rank = 0
        elif n_dev < local_world:
            raise SystemExit(f"bench.py: {local_world} ranks on this node but only {n_dev} HIP devices are visible "
                             f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    use_comm = world > 1 or (args.force_comm and not EMU)
    rdzv = commlib.Rendezvous(rank, world) if use_comm else None

    Q = args.quality
    if args.workload == "config5":
        W, H, S, NF = args.width or 3840, args.height or 2160, args.streams or 1, args.frames or 300
    else:
        W, H, S, NF = args.width or 1920, args.height or 1080, args.streams or 96, args.frames or GOP

    # ---- stream assignment: rank 0 decides, everyone learns it through one tiny broadcast (the only "scatter" this path
    # has: stream ids / seeds, a few hundred bytes); config5: seed = base + stream id = base + rank
    if args.workload == "config5":
        sid = np.arange(S * world, dtype=np.int64)
        table = np.stack([sid % world, pkg.synth.SEED + sid, sid], axis=1)
    else:
        table = shard.assign_streams(n_streams_total=S * world, world=world, base_seed=pkg.synth.SEED)
    ctx = pkg.Context(local_rank)
    if os.environ.get("PFV_BENCH_PENC_FORM"):        # A/B runs: the p-frame encoder's form (PFV_OPT_TILE_COMPACTION: 0 strips, 1 tile compaction, 2 split kernels)
        ctx.set_option(pkg._lib.PFV_OPT_TILE_COMPACTION, int(os.environ["PFV_BENCH_PENC_FORM"]))
    # ncclCommInitRank of 8 ranks on one node takes seconds; past 90 s every rank falls back to the socket backend together (comm.py)
    comm_ctx = ctx
    if EMU and os.environ.get("PFV_BENCH_FAKE_RCCL_FAILURE") == "1":
        # test-only (tests/test_sharding.py::test_rccl_fallback_is_loud): pretend every rank has its own GPU and ncclCommInitRank fails on rank 1
        import ctypes

        class _FailingRccl:
            def pfv_comm_unique_id(self, p):
                ctypes.memset(p, 7, 128)
                return 0

            def pfv_comm_init(self, c, r, w, uid, out):
                ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.c_void_p(0)
                return -2 if r == 1 else 0

            def pfv_last_error(self, c):
                return b"simulated ncclCommInitRank failure"

            def pfv_comm_destroy(self, h):
                return 0

        class _Shim:
            handle, _lib, keep_alive = ctx.handle, _FailingRccl(), False
        share, comm_ctx = False, _Shim()
    comm = commlib.Comm(comm_ctx, rdzv, use_rccl=not share, init_timeout=float(os.environ.get("PFV_RCCL_INIT_TIMEOUT", "90"))) if use_comm else None
    rccl_fallback = bool(use_comm and world > 1 and not share and comm.backend != "rccl")
    if rccl_fallback:
        # one GPU per rank and RCCL asked for, but the communicator did not come up on every rank: the job carries on over the rendezvous
        # sockets (same results; the collectives are a table broadcast, barriers and one reduction) -- and says so LOUDLY
        print(f"bench.py[rank {rank}]: RCCL FALLBACK -- control plane on TCP sockets instead of RCCL/xGMI: {comm.rccl_error}", file=sys.stderr, flush=True)
    if use_comm:
        table = comm.broadcast_array(np.asarray(table, dtype=np.int64) if rank == 0 else np.zeros_like(np.asarray(table, dtype=np.int64)))
    mine = shard.streams_of_rank(table, rank)
    assert len(mine) == S

    timer = Timer(ctx, None)
    gop_batched = args.workload == "config5" and not args.serial_gops
    if gop_batched:       # the GOPs of the stream are the slots of a launch; frames resident in display order
        ss = GopBatchSet(pkg, ctx, W, H, Q, [int(r[1]) for r in mine], NF)
    else:
        ss = StreamSet(pkg, ctx, W, H, Q, [int(r[1]) for r in mine], NF)  # synthetic input generated in HBM: [NF][S][frame_bytes]
    n_mb = ss.n_mb

    ev = {k: [] for k in BYTES_PER_MB}

    def on_launch(name=None, a=None, b=None):
        # HIP events on the kernels' own stream bracket every launch of the timed steps (one event = ~1 us of host time,
        # inside the timed region)
        if name is None:
            return timer.stamp()
        ev[name].append((a, b))
        return None

    def barrier():
        ctx.sync()
        if not EMU:
            ctx.device_sync()                         # hipDeviceSynchronize (what torch.cuda.synchronize() does)
        if use_comm:
            comm.barrier()                            # RCCL: a 1-element all-reduce on the context's stream + synchronise

    for _ in range(args.warmup):
        ss.step()
    timer.reserve(args.steps * 4 * min(NF, 2 * GOP) + 64)
    sections["setup"] = time.perf_counter() - t_start  # build check, library load, rendezvous, synthetic input generated in HBM, warm-up
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ss.step(on_launch=on_launch)
    barrier()
    el = time.perf_counter() - t0
    sections["timed"] = el
    if not args.no_verify:
        ss.verify()                                   # decoder output == encoder reconstruction, no bad motion vector
    coded_frac = ss.coded_fraction()
    kern_ms = {k: float(np.mean([timer.ms(a, b) for a, b in v])) for k, v in ev.items() if v}
    pe_ms = kern_ms.get("k_enc_pframe", float("nan"))

    t1 = time.perf_counter()
    ent = None if args.no_entropy else entropy_side(ss, timer, args)
    sections["entropy"] = time.perf_counter() - t1

    total_mb, el_max = float(args.steps) * NF * S * n_mb, el
    per_rank = None
    if use_comm:
        mine_mb = total_mb
        total_mb = float(comm.allreduce([total_mb], "sum")[0])        # whole-job macroblocks
        el_max = float(comm.allreduce([el], "max")[0])                # the slowest rank's time
        # every rank's own figures on rank 0's line (a straggler GPU must be visible): one slot per rank in a vector that is summed
        bus = "0000:00:00.0" if EMU else ctx.pci_bus_id()
        dom, b, df = bus.split(":")
        dv, fn = df.split(".")
        slot = np.zeros((4, world))
        slot[:, rank] = (mine_mb / el, el, local_rank, (int(dom, 16) << 16) | (int(b, 16) << 8) | (int(dv, 16) << 3) | int(fn, 16))
        g = comm.allreduce(slot.reshape(-1), "sum").reshape(4, world)
        per_rank = [{"rank": r, "macroblocks_per_s": float(g[0, r]), "seconds": float(g[1, r]), "device_ordinal": int(g[2, r]),
                     "pci_bus_id": "%04x:%02x:%02x.%x" % (int(g[3, r]) >> 16, (int(g[3, r]) >> 8) & 0xff, (int(g[3, r]) >> 3) & 0x1f, int(g[3, r]) & 7)}
                    for r in range(world)]

    if rank == 0:
        launch_mbs = ss.launch_streams * n_mb        # macroblocks of a full launch (GOP-batched: every GOP of the stream)
        achieved = launch_mbs * BYTES_PER_MB_PENC / (pe_ms * 1e-3) / 1e9
        traffic, n_valu, traffic_source, traffic_waves = traffic_from_profiles(S, W, H, Q)
        live = None
        t1 = time.perf_counter()
        if world == 1 and not EMU and not args.no_live_pmc:
            live = live_pmc(args, S, W, H, Q, NF)     # child processes on the same GPU (288 GB: room for their copy of the workload)
            if live:
                traffic, n_valu, traffic_waves = live["traffic_bytes"], live["valu_wave_instructions"], live["wavefronts"]
                traffic_source = ("measured in this run: three child runs of this workload under rocprofv3 --kernel-trace --pmc (FETCH_SIZE x 2 KiB + WRITE_SIZE KiB; "
                                  "SQ_INSTS_VALU, SQ_WAVES), means over the k_enc_pframe launches")
        sections["live_pmc"] = time.perf_counter() - t1
        valu = issue = None
        if n_valu:
            valu = {"wave_instructions_per_launch": n_valu, "simd_cycles_per_instruction": pe_ms * 1e-3 * GPU_CLOCK_HZ * N_SIMDS / n_valu,
                    "note": "SQ_INSTS_VALU (see traffic_source), this run's launch time, 1024 SIMDs x 2.4 GHz"}
            # the roof that binds before HBM does: VALU issue.  A wave64 instruction of the kernel's mix occupies its SIMD16 for 4 cycles
            # (4.1-4.4 measured, profiles/r02_ubench_valu_rates2.txt); the issue floor is the launch's instruction count at that rate
            # on all 1024 SIMDs with no stall at all
            n_waves = traffic_waves or (launch_mbs / 8)
            floor_us = n_valu * VALU_CYCLES_PER_INSTR / N_SIMDS / GPU_CLOCK_HZ * 1e6
            issue = {"bound": "valu-issue", "valu_per_wavefront": n_valu / n_waves, "cycles_per_instr": pe_ms * 1e-3 * GPU_CLOCK_HZ * N_SIMDS / n_valu,
                     "issue_floor_us": floor_us, "frac_of_issue_floor": floor_us / (pe_ms * 1e3),
                     "assumed_cycles_per_instr_at_the_floor": VALU_CYCLES_PER_INSTR, "simds": N_SIMDS, "clock_hz": GPU_CLOCK_HZ,
                     "source": traffic_source}
        if gop_batched:
            name = (f"config5: one {W}x{H} {NF}-frame GOP-{GOP} stream per GPU (seed = base + rank), encode+decode, GOP-batched: frame t of all "
                    f"{ss.n_gops} GOPs of the stream in one launch per frame operation ({ss.n_gops} x {n_mb} macroblocks per launch; same coefficients, "
                    f"packets and frames as the serial pass -- an i-frame never reads prev_frame, src/enc.rs:84-97); frames resident in display order")
        elif args.workload == "config5":
            name = f"config5: one {W}x{H} {NF}-frame GOP-{GOP} stream per GPU (seed = base + rank), encode+decode, one launch per frame operation (--serial-gops)"
        else:
            name = f"{W}x{H} YUV420 GOP-{GOP} encode+decode, {S} independent streams per GPU batched per launch"
        res = {
            "metric": "macroblocks/s (encode+decode) 1080p YUV420" if (W, H) == (1920, 1080) else f"macroblocks/s (encode+decode) {W}x{H} YUV420",
            "value": total_mb / el_max,
            "unit": "macroblocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32 (encoder transforms evaluated in exact f32; decoders i32)",
            "data": f"synthetic, generated on the device ({S} distinct integer-hash texture streams per GPU, seed per stream; quality {Q})",
            "rccl_ranks": world if comm is not None and comm.backend == "rccl" else 0,
            "rccl_fallback": rccl_fallback,       # true: every rank had its own GPU, RCCL was asked for and did NOT come up (control_plane.rccl_error says why)
            "launch": {"mode": ("torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else ("self-launch" if os.environ.get("PFV_RDZV_NONCE") else "env")) if world > 1 else "single process",
                       "world": world, "visible_gpus": None if EMU else int(pkg._lib.load().pfv_device_count()),
                       "same_code_path_for_every_n": "one main() for N = 1 and for a rank of N > 1 (torch-free process; StreamSet.step in the timed loop); at N > 1 the "
                                                     "table broadcast, the barriers and the counter reduction go through comm.py, at N = 1 they are no-ops; the "
                                                     "extra / cpu_baseline / live-PMC legs run on rank 0 at N = 1 only, after the timed region"},
            "control_plane": {"backend": comm.backend if comm is not None else None, "shared_gpu": bool(share and world > 1), "emulated": EMU,
                              "rccl_error": comm.rccl_error if comm is not None else None, "ranks": per_rank,
                              "collectives": "assignment-table broadcast + barriers + counter all-reduce only (no data-path collective); "
                                             "RCCL through libpfv_hip.so (pfv_comm_*), no torch in the process"},
            "config": {"workload": name, "streams_per_gpu": S, "frames_per_step": NF, "macroblocks_per_frame": n_mb, "quality": Q,
                       "slots_per_launch": ss.launch_streams, "gop_batched": bool(gop_batched),
                       "pframe_coded_fraction": round(coded_frac, 4), "parallelism": f"streams sharded over {world} GPU(s)"},
            # The roof that binds k_enc_pframe is VALU issue (roofline.issue), the HBM figures -- `achieved`, `peak`, `frac`,
            # `traffic`: algorithmic / measured bytes per launch against the 8 TB/s spec peak -- are what the contract asks for and stay as they were
            "roofline": {"bound": "valu-issue", "kernel": "k_enc_pframe", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bound_note": "achieved / peak / frac / traffic are the HBM roofline of the kernel (algorithmic bytes per launch over its mean HIP-event "
                                       "duration; PMC bytes); the kernel sits at `issue.frac_of_issue_floor` of its VALU-issue floor, which is the roof it "
                                       "actually runs into before HBM",
                         "algorithmic_bytes_per_launch": launch_mbs * BYTES_PER_MB_PENC,
                         "avg_launch_ms": pe_ms, "macroblocks_per_launch": launch_mbs,
                         "algorithmic_bytes_per_macroblock": BYTES_PER_MB_PENC, "valu": valu, "issue": issue},
        }
        # the whole step against both roofs: SURVEY.md section 8d's GOP-15 average of 2 277.3 algorithmic bytes per macroblock (encode 1 266.7 +
        # decode 1 010.7; the fused retframe crop is NOT in that figure) over the step's wall time; and the step's VALU floor = every codec
        # kernel's instruction count at 4.2 cycles per wave64 instruction on 1 024 SIMDs, where the counts were collected (live PMC passes)
        gop_n = min(GOP, NF)
        step_b_per_mb = (BYTES_PER_MB_SURVEY["k_enc_iframe"] + BYTES_PER_MB_SURVEY["k_dec_iframe"] +
                         (gop_n - 1) * (BYTES_PER_MB_SURVEY["k_enc_pframe"] + BYTES_PER_MB_SURVEY["k_dec_pframe"])) / gop_n
        step_gbs = (total_mb / world) / args.steps * step_b_per_mb / (el_max / args.steps) / 1e9
        valu_floor_ms = None
        if live and all(k in live.get("kernels", {}) for k in kern_ms):
            per_launch = {k: live["kernels"][k]["valu_wave_instructions"] for k in kern_ms}
            launches = {k: (NF // GOP + (1 if NF % GOP else 0)) if k.endswith("iframe") else NF - (NF // GOP + (1 if NF % GOP else 0)) for k in kern_ms}
            if gop_batched:
                launches = {k: 1 if k.endswith("iframe") else gop_n - 1 for k in kern_ms}
            valu_floor_ms = sum(per_launch[k] * launches[k] for k in kern_ms) * VALU_CYCLES_PER_INSTR / N_SIMDS / GPU_CLOCK_HZ * 1e3
        res["step_roofline"] = {"algorithmic_bytes_per_macroblock": step_b_per_mb, "algorithmic_bytes_per_step": (total_mb / world) / args.steps * step_b_per_mb,
                                "achieved_GBps": step_gbs, "peak_GBps": HBM_PEAK_GBS, "frac": step_gbs / HBM_PEAK_GBS,
                                "valu_issue_floor_ms": valu_floor_ms, "frac_of_valu_issue_floor": (valu_floor_ms / (el_max / args.steps * 1e3)) if valu_floor_ms else None,
                                "note": "whole step (every encode and decode launch of the pass) per GPU: SURVEY 8d algorithmic bytes over the step's wall time "
                                        "against 8 TB/s; valu_issue_floor_ms = the four codec kernels' SQ_INSTS_VALU (this run's PMC passes) x launches per step "
                                        "x 4.2 cycles / 1 024 SIMDs / 2.4 GHz -- null when the counters were not collected in this run"}
        res["pframe_encode"] = {"value": launch_mbs / (pe_ms * 1e-3), "unit": "macroblocks/s",
                                "note": "k_enc_pframe alone (motion search + residual DCT + closed-loop reconstruction), HIP-event time"}
        res["kernels"] = {k: {"avg_launch_ms": ms, "macroblocks_per_s": launch_mbs / (ms * 1e-3),
                              "algorithmic_GBps": launch_mbs * BYTES_PER_MB[k] / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": launch_mbs * BYTES_PER_MB[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_macroblock": BYTES_PER_MB[k],
                              "frac_of_hbm_peak_survey_bytes": launch_mbs * BYTES_PER_MB_SURVEY[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_macroblock_survey": BYTES_PER_MB_SURVEY[k]}
                          for k, ms in kern_ms.items()}
        res["kernels_note"] = ("frac_of_hbm_peak counts what a launch moves: SURVEY 8d's bytes + 256 B per macroblock for the retframe crop the decode kernels "
                               "fuse (src/dec.rs:195-197, 209-211); frac_of_hbm_peak_survey_bytes counts SURVEY 8d's figure alone (768 / 1 028 B for the decoders)")
        if ent:
            res["encode_to_payload"] = ent
        host_gop = ss.host_frames(0, min(GOP, NF)) if (world == 1 and not args.no_cpu_baseline) else None
        if world == 1 and not args.no_extra and not EMU:
            extra = {}
            if args.workload == "config5":
                t1 = time.perf_counter()
                seed0 = ss.seeds[0]
                if gop_batched:
                    batched_rate = ss.wall(2)
                    ss.close()                        # the serial figure needs the frames in its own layout
                    extra["config4"] = stream_4k_side(pkg, ctx, Q, seed0, n_frames=NF, gop_batched_rate=batched_rate) if (W, H) == (3840, 2160) else None
                else:
                    extra["config4"] = stream_4k_side(pkg, ctx, Q, seed0, ss=ss) if (W, H) == (3840, 2160) else None
                    ss.close()
                sections["extra.config4"] = time.perf_counter() - t1
            else:
                ss.close()                            # give the 4.5 GB of resident input back first
                t1 = time.perf_counter()
                extra["config2"], extra["config3"] = config23_side(pkg, ctx, timer, Q, with_cpu=not args.no_cpu_baseline)
                sections["extra.config2_config3"] = time.perf_counter() - t1
                for name, fn in (("low_motion", lambda: low_motion_side(pkg, ctx, timer, W, H, Q, [int(r[1]) for r in mine], NF, kern_ms, coded_frac)),
                                 ("single_stream", lambda: single_stream_side(pkg, ctx, Q)),
                                 ("lookahead", lambda: lookahead_side(pkg, ctx, Q)),
                                 ("batch_encoder_end_to_end", lambda: batch_encoder_side(pkg, ctx, Q)),
                                 ("config4", lambda: stream_4k_side(pkg, ctx, Q, pkg.synth.SEED))):
                    t1 = time.perf_counter()
                    extra[name] = fn()
                    sections["extra." + name] = time.perf_counter() - t1
            res["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            t1 = time.perf_counter()
            res["cpu_baseline"] = cpu_baseline(W, H, Q, host_gop, n_mb, budget_s=4.0 if EMU else 24.0)
            sections["cpu_baseline"] = time.perf_counter() - t1
            if res["cpu_baseline"].get("pframe_encode_value"):
                res["pframe_encode"]["vs_cpu_baseline"] = res["pframe_encode"]["value"] / res["cpu_baseline"]["pframe_encode_value"]
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None                # rank 0 at N = 1 only
        sections["total"] = time.perf_counter() - t_start
        res["sections_s"] = {k: round(v, 3) for k, v in sections.items()}
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(res), flush=True)
        if stdout_fd is not None:
            os.dup2(2, 1)

    ss.close()
    timer.close()
    stuck = False
    if use_comm:
        comm.barrier()
        comm.close()
        stuck = comm.stuck           # ncclCommInitRank never came back on this rank: the context stays, the process leaves the hard way
        rdzv.close()
    ctx.close()
    if stuck:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
