"""Multi-GPU sharding of independent streams.

Streams (separate Encoder / Decoder instances, src/enc.rs:12-26, src/dec.rs:15-28) share
nothing, so stream s simply runs on rank ``s % world``.  No pixel or coefficient ever crosses
GPUs; the only exchanges are control-plane: one broadcast of the assignment table and one
reduction of the per-rank counters.  On the GPU node they run on RCCL through the library
(comm.py / csrc/pfv_comm.hip, no torch in the process); the gloo world-2 CPU tests run the same two
exchanges over torch.distributed with helpers of their own (tests/libswitch.py).
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
