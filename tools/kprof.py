#!/usr/bin/env python3
"""Where a workgroup of k_enc_pframe spends its time: the bench's 96-stream p-frame encode on a library built with -DPFV_KPROF
(clock64 of thread 0 at the phase boundaries, one row per workgroup); prints mean shader-clock cycles between the marks.
    hipcc ... -DPFV_KPROF -o /tmp/libpfv_kprof.so pfv_capi.hip;  PFV_HIP_LIB=/tmp/libpfv_kprof.so python tools/kprof.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as g
pkg = g.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
ctx = pkg.Context(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, [pkg.synth.SEED + 17 * k for k in range(S)], bench.GOP)
lib = ctypes.CDLL(os.environ["PFV_HIP_LIB"])
NAMES = ["issue window DMA + source loads", "barrier (window complete)", "search step 8", "step 4", "step 2", "step 1",
         "patch fetch", "barrier (window released)", "residual+forward+quantise+store, half 0", "inverse+reconstruct, half 0", "half 1"]
n_rows = 1 << 16
rows = np.zeros((n_rows, 16), np.uint64)
acc = np.zeros(len(NAMES)); life = 0.0; n = 0
ss.enc.encode_iframe_dev(ss.frame_ptr(0), ss.coef)
for t in range(1, ss.n_frames):
    ss.enc.encode_pframe_dev(ss.frame_ptr(t), ss.mv, ss.has, ss.coef)
    ctx.sync()
    assert lib.pfv_debug_kprof(rows.ctypes.data_as(ctypes.c_void_p), n_rows) == 0
    live = rows[:, 11] > 0
    r = rows[live][:, :12].astype(np.int64)
    interior = np.all(np.diff(r, axis=1) >= 0, axis=1)          # boundary tiles skip nothing, but keep the rows sane
    d = np.diff(r[interior], axis=1)
    acc += d.mean(0); life += d.sum(1).mean(); n += 1
# phase relation of the workgroups that share a CU (last launch): start times per CU, sorted
hw, xcc = rows[live][:, 12].astype(np.int64), rows[live][:, 13].astype(np.int64)
cu = (xcc & 15) * 4096 + ((hw >> 8) & 0xff)                   # XCC | se_id, sh_id, cu_id
t0, t1 = rows[live][:, 0].astype(np.int64), rows[live][:, 11].astype(np.int64)
print("distinct CUs seen:", len(np.unique(cu)))
gaps = []
for c in np.unique(cu)[:4]:
    st = np.sort(t0[cu == c]); st = st - st[0]
    print(f"CU {c:#x}: {len(st)} workgroups; first starts (cycles): {st[:16].tolist()}")
for c in np.unique(cu):
    st = np.sort(t0[cu == c])
    gaps.append(np.diff(st))
gaps = np.concatenate(gaps)
life_mean = (t1 - t0).mean()
print(f"start-to-start gaps on a CU: mean {gaps.mean():.0f} cycles (lifetime {life_mean:.0f} / 5 = {life_mean / 5:.0f} if evenly staggered); "
      f"share of gaps < 1000 cycles: {100 * (gaps < 1000).mean():.0f} %, < 3000: {100 * (gaps < 3000).mean():.0f} %")
print(f"k_enc_pframe, {S} x 1080p, mean over {n} p-frames: workgroup lifetime {life / n:.0f} cycles")
for nm, c in zip(NAMES, acc / n):
    print(f"  {nm:45s} {c:8.0f}  {100 * c / (life / n):5.1f} %")
ss.close(); ctx.close()
