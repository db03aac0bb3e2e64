#!/usr/bin/env python3
"""First-contact readiness of the N > 1 bench: BOTH ways the driver may start it, on the CPU emulator (no GPU needed), N ranks.

    python tools/preflight_multi.py [--ranks 8] [--mode self|env|torchrun|all]

  self      `python bench.py --gpus N`: bench.py launches its own N ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / PFV_RDZV_NONCE).
  env       the environment a launcher hands to each rank -- RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT,
            TORCHELASTIC_RUN_ID -- set by hand for N processes, with MASTER_PORT OCCUPIED by a listener (as torchrun's store occupies it) and
            the first port of the rendezvous walk occupied too (another service on the node): the rendezvous must walk on.
  torchrun  the driver's own command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
            --master-port P bench.py --gpus N ...` (needs torch for the launcher only; a rank stays a torch-free process).

PFV_BENCH_EMU=1 makes every rank run the product's session path with the kernel sources on the CPU emulator, at a toy geometry; ranks
"share" the emulated device, so the control plane is the TCP backend of comm.py -- exactly what a one-GPU box would do.  On a node with N
GPUs drop PFV_BENCH_EMU and the same three commands run RCCL (`rccl_ranks` = N, `rccl_fallback` false).  Each mode must print ONE JSON
line from rank 0 with n_gpus = N and the whole-job macroblock count."""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
sys.path.insert(0, ROOT)
TOY = ["--steps", "1", "--warmup", "1", "--streams", "2", "--width", "64", "--height", "48", "--frames", "3", "--no-entropy"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_line(out: str, n: int, mode: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"{mode}: expected ONE JSON line, got {len(lines)}"
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and res["scaling"] == "weak", res
    assert abs(res["value"] * res["ms_per_step"] * 1e-3 - n * 2 * 3 * 20) < 1e-6, "whole-job macroblock count"
    assert [r["rank"] for r in res["control_plane"]["ranks"]] == list(range(n))
    assert res["rccl_fallback"] is False            # shared (emulated) device: TCP by design, not a fallback
    assert res["launch"]["world"] == n
    return res


def base_env():
    env = dict(os.environ, PFV_BENCH_EMU="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "PFV_RDZV_NONCE"):
        env.pop(k, None)
    return env


def mode_self(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), *TOY], env=base_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = check_line(r.stdout, n, "self")
    assert res["launch"]["mode"] == "self-launch"
    return res


def mode_env(n):
    import importlib.util
    spec = importlib.util.spec_from_file_location("pfv_comm_for_port", os.path.join(ROOT, "pretty-fast-video_amd", "comm.py"))
    comm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(comm)
    # MASTER_PORT occupied (the launcher's store), and so is the first port of the rendezvous walk derived from it
    store = socket.socket()
    store.bind(("127.0.0.1", 0))
    store.listen(1)
    port = store.getsockname()[1]
    squat = socket.socket()
    squat.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    squatting = True
    try:
        squat.bind(("127.0.0.1", comm.rendezvous_port(port)))
        squat.listen(1)
    except OSError:
        squatting = False                            # someone else already sits there: the walk is exercised all the same
    procs = []
    try:
        for rank in range(n):
            env = dict(base_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=f"preflight-{os.getpid()}")
            procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", str(n), *TOY], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=900) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        store.close()
        squat.close()
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}: rc {p.returncode}\n{se[-1500:]}"
        assert rank == 0 or not any(ln.startswith("{") for ln in so.splitlines()), "only rank 0 prints the line"
    res = check_line(outs[0][0], n, "env")
    assert res["launch"]["mode"] == "torchrun"       # TORCHELASTIC_RUN_ID in the environment
    res["_port_walk_forced"] = squatting
    return res


def mode_torchrun(n):
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           BENCH, "--gpus", str(n), *TOY]
    r = subprocess.run(cmd, env=base_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    res = check_line(r.stdout, n, "torchrun")
    assert res["launch"]["mode"] == "torchrun"
    return res


MODES = {"self": mode_self, "env": mode_env, "torchrun": mode_torchrun}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--mode", default="all", choices=["all", *MODES])
    a = ap.parse_args()
    for m in (MODES if a.mode == "all" else [a.mode]):
        res = MODES[m](a.ranks)
        print(f"preflight {m:9s} OK: n_gpus {res['n_gpus']}, backend {res['control_plane']['backend']}, launch.mode {res['launch']['mode']}, "
              f"value {res['value']:.0f} macroblocks/s" + (f", rendezvous port walk forced: {res['_port_walk_forced']}" if "_port_walk_forced" in res else ""), flush=True)
