"""Test / experiment infrastructure, NOT part of the product: point the package's ctypes loader at another build of the same C ABI.

The product loader (pretty-fast-video_amd/_lib.py) loads the in-tree libpfv_hip.so and nothing else.  The CPU-emulator build of the
same sources (tests/hipemu) that the non-GPU suite runs, and the variant builds of the A/B scripts under tools/ (ablations, earlier
commits), are switched in from here by replacing the loader's `lib_path` function -- from the outside.

    use(pkg, path, allow_missing=False)   load `path` from now on (allow_missing: a build of an EARLIER commit may lack newer symbols)
    reset(pkg)                            back to the in-tree library
    apply_from_env(pkg)                   tools only: PFV_HIP_LIB=<path> [PFV_HIP_LIB_OLDER=1] in the environment
"""
from __future__ import annotations

import ctypes
import os

_orig = {}


def use(pkg, path: str, allow_missing: bool = False):
    L = pkg._lib
    if "lib_path" not in _orig:
        _orig["lib_path"], _orig["signatures"] = L.lib_path, list(L.SIGNATURES)
    L.lib_path = lambda: path
    L.SIGNATURES = list(_orig["signatures"])
    if allow_missing:
        probe = ctypes.CDLL(path)
        L.SIGNATURES = [s for s in L.SIGNATURES if hasattr(probe, s[0])]
    L._lib = None
    return path


def reset(pkg):
    L = pkg._lib
    if "lib_path" in _orig:
        L.lib_path, L.SIGNATURES = _orig["lib_path"], list(_orig["signatures"])
    L._lib = None


def apply_from_env(pkg):
    path = os.environ.get("PFV_HIP_LIB")
    if path:
        use(pkg, path, allow_missing=os.environ.get("PFV_HIP_LIB_OLDER") == "1")
    return path


# ---- the two control-plane exchanges of the sharded job over a torch.distributed group (gloo in the CPU tests; the product's ranks
# use RCCL through the library, pretty-fast-video_amd/comm.py, and have no torch in the process)
def broadcast_table(table, rank: int, dist, device=None):
    """rank 0's table to everyone (a few hundred bytes)"""
    import numpy as np
    import torch
    t = torch.as_tensor(np.asarray(table, dtype=np.int64) if rank == 0 else np.zeros_like(np.asarray(table, dtype=np.int64)))
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def gather_counters(macroblocks: float, seconds: float, checksum: int, dist, device=None):
    """(sum of macroblocks, max of seconds, sum of checksums mod 2^40) over all ranks"""
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
