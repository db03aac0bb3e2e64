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
    pkg = g.load_package()
    from importlib import import_module
    shard = import_module("pretty_fast_video_amd.shard")
    from oracle_bind import Oracle
    table = shard.assign_streams(n_total, world, pkg.synth.SEED) if rank == 0 else np.zeros((n_total, 3), np.int64)
    table = shard.broadcast_table(table, rank, dist)
    mine = shard.streams_of_rank(table, rank)
    ora = Oracle()
    cs, mbs = 0, 0
    for _, seed, _sid in mine:
        c, m = _encode_stream(pkg, ora, seed, 48, 32, 2)
        cs, mbs = cs + c, mbs + m
    tot_mb, max_s, tot_cs = shard.gather_counters(mbs, 1.0 + rank, cs, dist)
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
