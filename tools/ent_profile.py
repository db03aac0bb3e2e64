#!/usr/bin/env python3
"""Where a workgroup of k_ent_scan / k_ent_pack spends its time: runs the bench's 96-stream GOP-15 encode with the entropy
stage on a library built with -DPFV_ENT_PROFILE (clock64 of lane 0 at the marks, one row per workgroup) and prints the
mean time between the marks per frame kind.
    hipcc ... -DPFV_ENT_PROFILE -o /tmp/libpfv_prof.so pfv_capi.hip;  PFV_HIP_LIB=/tmp/libpfv_prof.so python tools/ent_profile.py"""
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
enc = ss.enc
enc.enable_entropy(async_stream=False)
lib = ctypes.CDLL(os.environ["PFV_HIP_LIB"])
n_groups = S * ((ss.n_mb * 4 + 255) // 256)
assert n_groups <= 1 << 15
rows = np.zeros((n_groups, 16), np.uint64)
SCAN = ["issue loads, stage->LDS", "barrier", "bitmap+prefix", "barrier", "symbol walk", "tail sums", "barrier", "group outputs"]
PACK = ["loads -> LDS init", "barrier", "headers + symbol steps", "barrier", "flush"]
PACK_MARKS = [0, 1, 2, 5, 6, 7]
acc = {}
for t in range(ss.n_frames):
    f = ss.frame_ptr(t)
    if t % bench.GOP == 0:
        enc.encode_iframe_dev(f, ss.coef); enc.pack_iframe_dev(ss.coef)
    else:
        enc.encode_pframe_dev(f, ss.mv, ss.has, ss.coef); enc.pack_pframe_dev(ss.mv, ss.has, ss.coef)
    ctx.sync()
    kind = "i-frame" if t == 0 else ("p-frame after the scene jump" if t == 8 else "p-frames")
    for k, names in ((0, SCAN), (1, PACK)):
        assert lib.pfv_debug_ent_profile(k, rows.ctypes.data_as(ctypes.c_void_p), n_groups) == 0
        marks = PACK_MARKS if k == 1 else list(range(len(names) + 1))
        d = np.diff(rows[:, marks].astype(np.int64), axis=1)
        span = int(rows[:, marks[-1]].max() - rows[:, 0].min())
        a = acc.setdefault((kind, k), [0, np.zeros(len(names)), 0, 0])
        a[0] += 1; a[1] += d.mean(0); a[2] += span; a[3] += d.sum(1).mean()
for (kind, k), (n, d, span, life) in acc.items():
    names = SCAN if k == 0 else PACK
    print(f"{kind:30s} {'scan' if k == 0 else 'pack'}: kernel span {span / n:9.0f} ticks, workgroup lifetime {life / n:7.0f}:  " +
          "  ".join(f"{nm}={c:.0f}" for nm, c in zip(names, d / n)))
ss.close(); ctx.close()
