"""Control plane of the multi-GPU path: one process per GPU, no data-path collective.

Two layers:

``Rendezvous``  a TCP star on MASTER_ADDR (127.0.0.1 for the one-node jobs this is used for): rank 0 listens, the other ranks
                connect.  It carries the 128-byte ncclUniqueId to the ranks and doubles as the whole control plane where RCCL
                cannot run: ranks sharing one GPU (developer dry run on a one-GPU box; RCCL refuses two ranks on one device)
                and the CPU-emulator tests.
``Comm``        the collectives the job needs -- broadcast of the assignment table, sum / max of the counters, barrier -- on
                RCCL through the library (pfv_comm_*, csrc/pfv_comm.hip: device buffers, the context's own HIP stream, xGMI)
                when every rank has its own GPU, on the rendezvous sockets otherwise.  ``backend`` says which ("rccl" | "tcp").

No torch: a rank process keeps a single HIP runtime (torch's wheel bundles its own, and two runtimes in one process double
the host cost of every small launch).
"""
from __future__ import annotations

import ctypes
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"PFVRDZV1"


def rendezvous_port(master_port: int) -> int:
    """a port of our own next to the launcher's (torchrun's store owns MASTER_PORT itself)"""
    return 20000 + (int(master_port) * 7 + 7919) % 40000


class Rendezvous:
    def __init__(self, rank: int, world: int, addr: str = None, port: int = None, timeout: float = 300.0):
        self.rank, self.world = int(rank), int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = rendezvous_port(int(port if port is not None else os.environ.get("MASTER_PORT", "29500")))
        # the job's token: the launcher's per-job nonce when there is one (bench.py's self-launcher sets PFV_RDZV_NONCE, torchrun a run id),
        # so that a process of another job on the node cannot take a rank's place
        nonce = (os.environ.get("PFV_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or "").encode()[:32].ljust(32, b"\0")
        token = _MAGIC + struct.pack("<ii", base, self.world) + nonce
        self.peers = {}           # rank 0: {rank: socket}
        self.sock = None          # other ranks: socket to rank 0
        self.listener = None
        if self.world == 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            for k in range(16):   # a busy port: walk on; the others try the same sequence and check the token
                try:
                    ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    ls.bind((addr, base + k))
                    break
                except OSError:
                    ls.close()
                    ls = None
            if ls is None:
                raise RuntimeError("rendezvous: no free port")
            ls.listen(self.world)
            ls.settimeout(timeout)
            self.listener = ls
            while len(self.peers) < self.world - 1:
                c, _ = ls.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(5.0)                     # a stray connection must not stall the job
                try:
                    hello = self._recv_exact(c, len(token) + 4)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                if hello[:len(token)] != token:
                    c.close()
                    continue
                r = struct.unpack("<i", hello[len(token):])[0]
                if not 1 <= r < self.world:            # not a rank of this job
                    c.close()
                    continue
                try:
                    c.sendall(b"OK")
                    if self._recv_exact(c, 2) != b"GO":   # the rank confirms it is still there (it may have given up on this
                        raise ConnectionError             # connection while it sat in the backlog, and come back on a new one)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                c.settimeout(timeout)
                if r in self.peers:                    # the rank came back on a new connection after giving up on one that sat in the
                    self.peers[r].close()              # backlog (it only ever completes the GO step on the connection it keeps)
                self.peers[r] = c
        else:
            while True:
                for k in range(16):
                    c = None
                    try:
                        c = socket.create_connection((addr, base + k), timeout=2.0)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(10.0)            # whoever listens there may not be rank 0 (the walk passes other services' ports)
                        c.sendall(token + struct.pack("<i", self.rank))
                        if self._recv_exact(c, 2) == b"OK":
                            c.sendall(b"GO")
                            c.settimeout(timeout)
                            self.sock = c
                            break
                        c.close()
                    except (OSError, ConnectionError):
                        if c is not None:
                            c.close()
                        continue
                if self.sock is not None:
                    break
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank 0 not reachable")
                time.sleep(0.05)

    @staticmethod
    def _recv_exact(s, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("rendezvous: peer closed")
            buf += chunk
        return bytes(buf)

    def _send_msg(self, s, b: bytes):
        s.sendall(struct.pack("<I", len(b)) + b)

    def _recv_msg(self, s) -> bytes:
        n = struct.unpack("<I", self._recv_exact(s, 4))[0]
        return self._recv_exact(s, n)

    def bcast(self, b: bytes = None) -> bytes:
        """rank 0's bytes to everyone"""
        if self.world == 1:
            return b
        if self.rank == 0:
            for r in sorted(self.peers):
                self._send_msg(self.peers[r], b)
            return b
        return self._recv_msg(self.sock)

    def gather(self, b: bytes):
        """every rank's bytes, in rank order, on rank 0 (None elsewhere)"""
        if self.world == 1:
            return [b]
        if self.rank == 0:
            out = {0: b}
            for r in sorted(self.peers):
                out[r] = self._recv_msg(self.peers[r])
            return [out[r] for r in range(self.world)]
        self._send_msg(self.sock, b)
        return None

    def allgather(self, b: bytes):
        parts = self.gather(b)
        blob = self.bcast(b"".join(struct.pack("<I", len(p)) + p for p in parts) if self.rank == 0 else None)
        out, pos = [], 0
        while pos < len(blob):
            n = struct.unpack_from("<I", blob, pos)[0]
            out.append(blob[pos + 4:pos + 4 + n])
            pos += 4 + n
        return out

    def barrier(self):
        self.allgather(b"")

    def close(self):
        for s in list(self.peers.values()) + [self.sock, self.listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.sock, self.listener = {}, None, None


class Comm:
    """broadcast / all-reduce / barrier for the job: RCCL on device buffers when `use_rccl`, the rendezvous sockets otherwise"""

    def __init__(self, ctx, rdzv: Rendezvous, use_rccl: bool, init_timeout: float = 180.0):
        self.ctx, self.rdzv, self.rank, self.world = ctx, rdzv, rdzv.rank, rdzv.world
        self.handle = None
        self.backend = "tcp"
        self.rccl_error = None        # why RCCL is not in use although it was asked for
        self._init_thread, self._init_box = None, None   # a pfv_comm_init that did not come back in time (see close())
        if use_rccl:
            self._init_rccl(init_timeout)

    def _init_rccl(self, timeout: float):
        """ncclCommInitRank is collective and blocks until every rank has joined; if it fails or does not come back on ANY rank,
        ALL ranks carry on with the socket backend (the job only needs a broadcast, a few barriers and one reduction) and the
        bench line says so (`rccl_ranks` 0, `control_plane.rccl_error`) instead of hanging the node."""
        import threading
        lib, ctx = self.ctx._lib, self.ctx
        uid = np.zeros(128, np.uint8)
        err = None
        if self.rank == 0:
            rc = lib.pfv_comm_unique_id(uid.ctypes.data_as(ctypes.c_void_p))
            if rc != 0:
                err = f"pfv_comm_unique_id: {rc} {(lib.pfv_last_error(None) or b'').decode()}"
        uid = np.frombuffer(self.rdzv.bcast(uid.tobytes() if self.rank == 0 else None), np.uint8).copy()
        box = {}

        def run():
            h = ctypes.c_void_p()
            box["rc"] = lib.pfv_comm_init(ctx.handle, self.rank, self.world, uid.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h))
            box["h"] = h
        if err is None and uid.any():
            t = threading.Thread(target=run, daemon=True)
            t.start()
            t.join(timeout)
            if t.is_alive():
                err = f"pfv_comm_init did not return within {timeout:.0f} s"
                # the thread is still INSIDE the library with this context: the context must outlive it (close() below)
                self._init_thread, self._init_box = t, box
                ctx.keep_alive = True
            elif box["rc"] != 0:
                err = f"pfv_comm_init: {box['rc']} {(lib.pfv_last_error(ctx.handle) or b'').decode()}"
        elif err is None:
            err = "rank 0 could not create the ncclUniqueId"
        # every rank learns whether all of them made it
        verdicts = self.rdzv.allgather((err or "").encode())
        bad = [f"rank {r}: {v.decode()}" for r, v in enumerate(verdicts) if v]
        if bad:
            self.rccl_error = "; ".join(bad)[:500]
            if err is None and box.get("h") is not None and box["h"].value:
                lib.pfv_comm_destroy(box["h"])
            return
        self.handle = box["h"]
        self.backend = "rccl"

    def broadcast_array(self, a: np.ndarray, root: int = 0) -> np.ndarray:
        """root's array (same shape and dtype on every rank) to everyone"""
        a = np.ascontiguousarray(a)
        if self.handle is None:
            assert root == 0
            return np.frombuffer(self.rdzv.bcast(a.tobytes() if self.rank == 0 else None), a.dtype).reshape(a.shape).copy()
        dev = self.ctx.alloc(max(a.nbytes, 16))
        try:
            self.ctx.upload(dev, a)
            self.ctx.check(self.ctx._lib.pfv_comm_broadcast_dev(self.handle, ctypes.c_void_p(dev), a.nbytes, int(root)))
            out = np.empty_like(a)
            self.ctx.download(out, dev)          # synchronises the stream
        finally:
            self.ctx.free(dev)
        return out

    def allreduce(self, values, op: str) -> np.ndarray:
        """element-wise "sum" or "max" of float64 values over the ranks"""
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        if self.handle is None:
            parts = [np.frombuffer(p, np.float64) for p in self.rdzv.allgather(v.tobytes())]
            return np.sum(parts, axis=0) if op == "sum" else np.max(parts, axis=0)
        out = v.copy()
        for at in range(0, out.size, 64):       # the host-value form stages at most 64 values per call
            part = out[at:at + 64]
            self.ctx.check(self.ctx._lib.pfv_comm_allreduce_f64(self.handle, part.ctypes.data_as(ctypes.c_void_p), part.size, 0 if op == "sum" else 1))
        return out

    def barrier(self):
        if self.handle is None:
            self.rdzv.barrier()
        else:
            self.ctx.check(self.ctx._lib.pfv_comm_barrier(self.handle))

    @property
    def stuck(self) -> bool:
        """a pfv_comm_init call is still inside the library (ncclCommInitRank never returned): the context cannot be destroyed under it
        and the process should leave through os._exit once its output is written"""
        return self._init_thread is not None and self._init_thread.is_alive()

    def close(self):
        if self._init_thread is not None:
            self._init_thread.join(1.0)           # it may have come back late
            if not self._init_thread.is_alive():
                h = self._init_box.get("h")
                if self._init_box.get("rc") == 0 and h is not None and h.value and self.ctx.handle:
                    self.ctx._lib.pfv_comm_destroy(h)   # the late communicator and its scratch buffer
                self._init_thread, self._init_box = None, None
                self.ctx.keep_alive = False
        if self.handle is not None and self.ctx.handle:
            self.ctx._lib.pfv_comm_destroy(self.handle)
        self.handle = None
