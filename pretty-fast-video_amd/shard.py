"""Multi-GPU sharding of independent streams.

Streams (separate Encoder / Decoder instances, src/enc.rs:12-26, src/dec.rs:15-28) share
nothing, so stream s simply runs on rank ``s % world``.  No pixel or coefficient ever crosses
GPUs; the only exchanges are control-plane: one broadcast of the assignment table and one
reduction of the per-rank counters.  On the GPU node they run on RCCL through the library
(comm.py / csrc/pfv_comm.hip, no torch in the process); ``broadcast_table`` / ``gather_counters``
below are the same two exchanges over a torch.distributed group -- gloo in the CPU tests
(tests/test_sharding.py).
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


def assign_gops(n_frames: int, gop: int, world: int) -> np.ndarray:
    """GOP-level sharding of ONE stream (granularity 2): an i-frame never reads prev_frame and overwrites
    every plane (src/enc.rs:84-97), so each I P P ... run is independent.  int64 table [n_gops, 4] of
    (rank, gop_index, first_frame, n_frames_in_gop); GOP g -> rank g % world."""
    assert n_frames >= 0 and gop >= 1 and world >= 1
    n_gops = (n_frames + gop - 1) // gop
    g = np.arange(n_gops, dtype=np.int64)
    first = g * gop
    return np.stack([g % world, g, first, np.minimum(gop, n_frames - first)], axis=1)


HEADER_BYTES = 20 + 4 * 128      # magic, version, w, h, fps, n_qtables, four 64 x u16 tables (src/enc.rs:190-219)


def encode_gops(pkg, ctx, frame_of, width: int, height: int, framerate: int, quality: int, rows) -> list:
    """Encode the GOPs listed in `rows` (rows of assign_gops owned by this rank) with fresh Encoders; returns
    [(gop_index, packet bytes)] -- the stream's packets for those frames, without header and EOF."""
    import io
    out = []
    for _, g, first, n in np.asarray(rows, dtype=np.int64):
        buf = io.BytesIO()
        enc = pkg.Encoder(buf, width, height, framerate, quality, ctx)
        for t in range(int(first), int(first + n)):
            (enc.encode_iframe if t == first else enc.encode_pframe)(frame_of(t))
        data = buf.getvalue()            # header + packets so far; finish() would append the EOF packet
        enc.finished = True              # the packets are spliced into a longer stream: no EOF here
        enc.close()
        out.append((int(g), data[HEADER_BYTES:]))
    return out


def splice_stream(header: bytes, gop_packets) -> bytes:
    """header + the GOPs' packets in order + EOF packet (src/enc.rs:221-227)"""
    body = b"".join(p for _, p in sorted(gop_packets))
    return header + body + bytes([0, 0, 0, 0, 0])


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
