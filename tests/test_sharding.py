"""N > 1 path on CPU: world_size-2 gloo run of the stream sharding + control-plane exchange.
Each rank encodes ITS streams with the CPU oracle standing in for the per-GPU hot path (the
kernels need a GPU); the test checks that the union over ranks equals the single-process
result -- i.e. sharding by stream changes nothing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checksum(arrs):
    return sum(int(np.asarray(a).astype(np.int64).sum()) for a in arrs)


def _encode_stream(pkg, oracle, seed, w, h, n_frames):
    st = pkg.SyntheticStream(w, h, seed=int(seed))
    enc = oracle.encoder(w, h, 5)
    cs = 0
    for t in range(n_frames):
        if t == 0:
            cs += _checksum([enc.encode_iframe(st.frame(t))])
        else:
            cs += _checksum(enc.encode_pframe(st.frame(t)))
    return cs + _checksum([enc.prev_frame()]), n_frames * enc.total_blocks


def _worker(rank, world, port, n_total, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    import libswitch
    pkg = g.load_package()
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    from oracle_bind import Oracle
    table = shard.assign_streams(n_total, world, pkg.synth.SEED) if rank == 0 else np.zeros((n_total, 3), np.int64)
    table = libswitch.broadcast_table(table, rank, dist)
    mine = shard.streams_of_rank(table, rank)
    ora = Oracle()
    cs, mbs = 0, 0
    for _, seed, _sid in mine:
        c, m = _encode_stream(pkg, ora, seed, 48, 32, 2)
        cs, mbs = cs + c, mbs + m
    tot_mb, max_s, tot_cs = libswitch.gather_counters(mbs, 1.0 + rank, cs, dist)
    if rank == 0:
        q.put((tot_mb, max_s, tot_cs, [int(x) for x in mine[:, 2]]))
    dist.destroy_process_group()


def test_stream_sharding_gloo_world2(pkg, oracle):
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    n_total, world = 5, 2
    table = shard.assign_streams(n_total, world, pkg.synth.SEED)
    assert sorted(table[:, 2]) == list(range(n_total))
    assert [len(shard.streams_of_rank(table, r)) for r in range(world)] == [3, 2]
    # single-process reference
    ref_cs, ref_mb = 0, 0
    for _, seed, _ in table:
        c, m = _encode_stream(pkg, oracle, seed, 48, 32, 2)
        ref_cs, ref_mb = ref_cs + c, ref_mb + m
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    tot_mb, max_s, tot_cs, rank0_ids = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tot_mb == ref_mb
    assert max_s == 2.0                      # max over ranks
    assert tot_cs == ref_cs % (1 << 40) or tot_cs == ref_cs   # checksum of checksums
    assert rank0_ids == [0, 2, 4]


def _gop_worker(rank, world, port, emu_lib, geom, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    import libswitch
    pkg = g.load_package()
    libswitch.use(pkg, emu_lib)                 # the product's own sources on the CPU emulator (kernels need a GPU)
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    import stream_cases as sc
    w, h, fps, quality, n_frames, gop = geom
    table = shard.assign_gops(n_frames, gop, world) if rank == 0 else np.zeros(((n_frames + gop - 1) // gop, 4), np.int64)
    table = libswitch.broadcast_table(table, rank, dist)          # the only scatter: frame index ranges
    st = pkg.SyntheticStream(w, h)
    with pkg.Context(0) as ctx:
        mine = shard.encode_gops(pkg, ctx, lambda t: sc.frame_of(pkg, w, h, st.frame(t)), w, h, fps, quality, table[table[:, 0] == rank])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)                  # packets are host bytes: concatenated in order on rank 0
    if rank == 0:
        q.put([x for part in gathered for x in part])
    dist.destroy_process_group()


def test_gop_sharding_gloo_world2(pkg, oracle):
    """one stream, GOPs dealt to two ranks, packets spliced on rank 0 == the single-process stream == the oracle's"""
    import io
    from importlib import import_module
    import conftest
    import stream_cases as sc
    from oracle_bind import OracleStreamEncoder
    shard = import_module("pretty_fast_video_amd.shard")
    geom = (48, 32, 30, 5, 8, 3)                               # 8 frames, GOP 3 -> GOPs of 3, 3, 2 frames
    w, h, fps, quality, n_frames, gop = geom
    table = shard.assign_gops(n_frames, gop, 2)
    assert table.tolist() == [[0, 0, 0, 3], [1, 1, 3, 3], [0, 2, 6, 2]]
    emu_lib = conftest.build_emulator()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gop_worker, args=(r, 2, port, emu_lib, geom, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g for g, _ in parts) == [0, 1, 2]
    # the oracle's stream of the same frames with the same i-frame positions
    st = pkg.SyntheticStream(w, h)
    oenc = OracleStreamEncoder(oracle, w, h, fps, quality)
    for t in range(n_frames):
        (oenc.encode_iframe if t % gop == 0 else oenc.encode_pframe)(st.frame(t))
    oenc.finish()
    want = oenc.bytes()
    assert shard.splice_stream(want[:shard.HEADER_BYTES], parts) == want


def _comm_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import __graft_entry__ as g
    g.load_package()
    from importlib import import_module
    commlib = import_module("pretty_fast_video_amd.comm")
    rdzv = commlib.Rendezvous(rank, world)
    got = rdzv.bcast(b"table-bytes" if rank == 0 else None)
    parts = rdzv.gather(bytes([rank]) * (rank + 1))
    allp = rdzv.allgather(b"r%d" % rank)
    rdzv.barrier()
    comm = commlib.Comm(None, rdzv, use_rccl=False)                 # the socket backend of the same collectives
    tab = comm.broadcast_array(np.arange(12, dtype=np.int64).reshape(4, 3) * (rank == 0))
    s = comm.allreduce([rank + 1.0, 10.0 * rank], "sum")
    m = comm.allreduce([rank + 1.0, 10.0 * rank], "max")
    comm.barrier()
    q.put((rank, got, parts, allp, tab.tolist(), s.tolist(), m.tolist(), comm.backend))
    rdzv.close()


def test_tcp_rendezvous_and_socket_collectives_world3():
    """comm.py without a GPU: the rendezvous that carries the ncclUniqueId, and the socket backend of broadcast / all-reduce /
    barrier that stands in for RCCL when ranks share a device"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, parts, allp, tab, s, m, backend in res:
        assert got == b"table-bytes" and allp == [b"r0", b"r1", b"r2"] and backend == "tcp"
        assert parts == ([b"\x00", b"\x01\x01", b"\x02\x02\x02"] if rank == 0 else None)
        assert tab == (np.arange(12).reshape(4, 3)).tolist()
        assert s == [6.0, 30.0] and m == [3.0, 20.0]


def _fallback_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes
    import __graft_entry__ as g
    g.load_package()
    from importlib import import_module
    commlib = import_module("pretty_fast_video_amd.comm")

    class FakeLib:                      # stands in for libpfv_hip.so: rank 1's ncclCommInitRank "fails"
        destroyed = 0

        def pfv_comm_unique_id(self, p):
            ctypes.memset(p, 7, 128)
            return 0

        def pfv_comm_init(self, ctx, r, w, uid, out):
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.c_void_p(0x1234)
            return -2 if r == 1 else 0

        def pfv_last_error(self, ctx):
            return b"ncclCommInitRank: unhandled system error"

        def pfv_comm_destroy(self, h):
            FakeLib.destroyed += 1

    class FakeCtx:
        handle = ctypes.c_void_p(1)
        _lib = FakeLib()

    rdzv = commlib.Rendezvous(rank, world)
    comm = commlib.Comm(FakeCtx(), rdzv, use_rccl=True, init_timeout=30.0)
    s = comm.allreduce([1.0 + rank], "sum")
    comm.barrier()
    q.put((rank, comm.backend, comm.rccl_error, s.tolist(), FakeLib.destroyed))
    rdzv.close()


def test_rccl_init_failure_on_one_rank_sends_every_rank_to_the_socket_backend():
    """ncclCommInitRank failing (or hanging) on any rank must not take the job down: all ranks agree over the rendezvous, the one
    that had succeeded destroys its communicator, everyone carries on with the socket collectives and reports why"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, backend, err, s, destroyed in res:
        assert backend == "tcp" and "rank 1" in err and "unhandled system error" in err and s == [3.0]
        assert destroyed == (1 if rank == 0 else 0)


def test_rccl_init_that_hangs_keeps_the_context_alive_under_it():
    """the init watchdog gives up on a pfv_comm_init that does not come back, but the thread is still INSIDE the library
    with the context: the context must not be destroyed under it (Context.close() becomes a no-op, the process is expected to leave
    through os._exit), and a call that returns late leaves a communicator that close() destroys"""
    import ctypes
    import time
    import __graft_entry__ as g
    g.load_package()
    from importlib import import_module
    commlib = import_module("pretty_fast_video_amd.comm")

    class FakeLib:
        destroyed = 0

        def pfv_comm_unique_id(self, p):
            ctypes.memset(p, 7, 128)
            return 0

        def pfv_comm_init(self, ctx, r, w, uid, out):
            time.sleep(0.8)                                  # ncclCommInitRank stuck on a dead peer
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.c_void_p(0x1234)
            return 0

        def pfv_last_error(self, ctx):
            return b""

        def pfv_comm_destroy(self, h):
            FakeLib.destroyed += 1

    class FakeCtx:
        handle = ctypes.c_void_p(1)
        _lib = FakeLib()
        keep_alive = False

    ctx = FakeCtx()
    comm = commlib.Comm(ctx, commlib.Rendezvous(0, 1), use_rccl=True, init_timeout=0.1)
    assert comm.backend == "tcp" and "did not return" in comm.rccl_error
    assert comm.stuck and ctx.keep_alive is True             # the context stays while the call is in it
    time.sleep(1.0)
    comm.close()                                             # the call came back late: its communicator is destroyed, the context is free again
    assert not comm.stuck and ctx.keep_alive is False and FakeLib.destroyed == 1


def _run_bench(extra_args, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ, PFV_BENCH_EMU="1", **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra_args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run, both ranks
    run the product's session path (kernel sources on the CPU emulator; PFV_BENCH_EMU=1 is test-only) on their own streams,
    the control plane (table broadcast, barriers, counter reduction) runs on the TCP rendezvous of comm.py (RCCL needs a GPU per
    rank), rank 0 prints the line."""
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--streams", "2", "--width", "64", "--height", "48", "--frames", "3",
                      "--no-entropy"])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["control_plane"]["backend"] == "tcp"
    assert res["rccl_ranks"] == 0 and res["control_plane"]["emulated"] is True
    # whole-job count: 2 ranks x 2 streams x 3 frames x 20 macroblocks per step
    assert abs(res["value"] * res["ms_per_step"] * 1e-3 - 2 * 2 * 3 * 20) < 1e-6
    assert res["cpu_baseline"] is None and "roofline" in res and res["config"]["streams_per_gpu"] == 2
    # every rank's own rate, device and PCI address on rank 0's line (a straggler must be visible)
    ranks = res["control_plane"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["macroblocks_per_s"] > 0 and r["seconds"] > 0 and r["pci_bus_id"] for r in ranks)
    assert abs(sum(r["macroblocks_per_s"] * r["seconds"] for r in ranks) - 2 * 2 * 3 * 20) < 1e-6


def test_bench_config5_single_rank_fields():
    """config-5 workload (one stream per GPU, seed = base + rank) and the cpu_baseline fields, at a toy geometry"""
    res = _run_bench(["--workload", "config5", "--steps", "1", "--warmup", "0", "--width", "64", "--height", "48", "--frames", "19",
                      "--no-entropy"])
    assert res["n_gpus"] == 1 and res["config"]["workload"].startswith("config5") and res["config"]["streams_per_gpu"] == 1
    # GOP-batched by default: the 15 + 4 frames are two GOPs = two slots per launch; the whole-job count still is frames x macroblocks
    assert res["config"]["gop_batched"] is True and res["config"]["slots_per_launch"] == 2
    assert abs(res["value"] * res["ms_per_step"] * 1e-3 - 19 * 20) < 1e-6
    cb = res["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "host_cpus", "cgroup_cpu_quota", "affinity_cpus", "value_1thread", "value_best",
              "threads_best"):
        assert k in cb, k
    # `cores` is a core count -- the CPUs' worth of time the container owns -- not a pool size; the pool that won is threads_best, one of
    # {1, usable, 2 x usable}; the value is the median of three passes of the winner
    assert cb["kind"] == "port" and cb["cores"] == cb["usable_cpus"] and "threads" not in cb and cb["value_1thread"] > 0
    quota = cb["cgroup_cpu_quota"]
    assert quota in (None, "max") or cb["cores"] <= max(1, round(float(quota)))
    assert cb["threads_best"] in (1, cb["usable_cpus"], 2 * cb["usable_cpus"], cb["affinity_cpus"])
    assert len(cb["passes_of_the_winner"]) == 3 and cb["passes_of_the_winner"][0] <= cb["value"] <= cb["passes_of_the_winner"][2] + 1
    assert cb["value_best"] >= 0.6 * cb["value_1thread"]
    assert set(res["sections_s"]) >= {"setup", "timed", "cpu_baseline"} and res["step_roofline"]["algorithmic_bytes_per_macroblock"] > 2000


@pytest.mark.parametrize("mode", ["self", "env", "torchrun"])
def test_preflight_eight_ranks(mode):
    """First-contact readiness of the 8-GPU run (tools/preflight_multi.py): `bench.py --gpus 8` in the three ways it can be started -- its own
    launcher; the launcher environment set by hand (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / TORCHELASTIC_RUN_ID) with MASTER_PORT and the
    first port of the rendezvous walk OCCUPIED; and the driver's command line (python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...) -- 8 ranks on the CPU emulator: one JSON line from rank 0, whole-job
    count, every rank's own row, `rccl_fallback` false (a shared device is TCP by design), `launch.mode` as started."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import preflight_multi as pm
    res = pm.MODES[mode](8)
    assert res["n_gpus"] == 8 and res["control_plane"]["backend"] == "tcp" and res["rccl_ranks"] == 0
    assert res["launch"]["mode"] == ("self-launch" if mode == "self" else "torchrun")
    if mode == "env":
        assert res["_port_walk_forced"] in (True, False)


def test_rccl_fallback_is_loud(tmp_path):
    """one GPU per rank, RCCL asked for, ncclCommInitRank failing on a rank: the job finishes over the sockets, says `rccl_fallback`: true at
    the TOP level of the line and prints the reason on stderr (bench.py; PFV_BENCH_FAKE_RCCL_FAILURE is a test hook of the emulator run)"""
    import json
    import subprocess
    env = dict(os.environ, PFV_BENCH_EMU="1", PFV_BENCH_FAKE_RCCL_FAILURE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--streams", "1", "--width", "64", "--height", "48",
                        "--frames", "2", "--no-entropy"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["rccl_fallback"] is True and res["rccl_ranks"] == 0 and res["control_plane"]["backend"] == "tcp"
    assert "simulated" in res["control_plane"]["rccl_error"]
    assert "RCCL FALLBACK" in r.stderr and "simulated" in r.stderr
