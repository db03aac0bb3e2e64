"""Multi-GPU sharding of independent streams.

Streams (separate Encoder / Decoder instances, src/enc.rs:12-26, src/dec.rs:15-28) share
nothing, so stream s simply runs on rank ``s % world``.  No pixel or coefficient ever crosses
GPUs; the only exchanges are control-plane: one broadcast of the assignment table and one
reduction of the per-rank counters.  Works with any torch.distributed backend (RCCL on the
GPU node, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def assign_streams(n_streams_total: int, world: int, base_seed: int) -> np.ndarray:
    """int64 table [n_streams_total, 3] of (rank, seed, stream_id); stream s -> rank s % world"""
    assert n_streams_total >= 0 and world >= 1
    sid = np.arange(n_streams_total, dtype=np.int64)
    return np.stack([sid % world, base_seed + 17 * sid, sid], axis=1)


def streams_of_rank(table: np.ndarray, rank: int) -> np.ndarray:
    """rows of the table owned by `rank`, in stream-id order"""
    table = np.asarray(table)
    return table[table[:, 0] == rank]


def broadcast_table(table, rank: int, dist, device=None) -> np.ndarray:
    """rank 0's table to everyone (a few hundred bytes)"""
    import torch
    t = torch.as_tensor(np.asarray(table, dtype=np.int64) if rank == 0 else np.zeros_like(np.asarray(table, dtype=np.int64)))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def gather_counters(macroblocks: float, seconds: float, checksum: int, dist, device=None):
    """(sum of macroblocks, max of seconds, xor-free sum of checksums mod 2^62) over all ranks"""
    import torch
    a = torch.tensor([float(macroblocks)], dtype=torch.float64)
    b = torch.tensor([float(seconds)], dtype=torch.float64)
    c = torch.tensor([int(checksum) % (1 << 40)], dtype=torch.int64)
    if device is not None:
        a, b, c = a.to(device), b.to(device), c.to(device)
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    dist.all_reduce(b, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(a.item()), float(b.item()), int(c.item())
