#!/usr/bin/env python3
"""When could NO resident wavefront of a SIMD issue?  k_enc_pframe on a library built with -DPFV_KPROF=2 (one row of clock64 stamps per
WAVEFRONT at the phase boundaries + HW_ID / XCC_ID): the bench's 96-stream p-frame launch, then per SIMD the timeline of its resident
wavefronts is rebuilt and every cycle of the launch is classified by how many wavefronts were resident and how many of those were
RUNNABLE, i.e. not parked in one of the kernel's three waits:
    fill     window DMA + source loads in flight, up to the first workgroup barrier (s_waitcnt vmcnt(0) + s_barrier)
    release  the second workgroup barrier (window released, strip masks published)
(everything else -- search, transform -- counts as runnable; the LDS round trips inside them are not visible to the stamps).
    hipcc ... -DPFV_KPROF=2 -o /tmp/libpfv_kprof2.so pfv_capi.hip;  PFV_HIP_LIB=/tmp/libpfv_kprof2.so python tools/kprof_simd.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, __graft_entry__ as g
import libswitch
pkg = g.load_package()
libswitch.apply_from_env(pkg)
ctx = pkg.Context(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, [pkg.synth.SEED + 17 * k for k in range(S)], 3)
lib = ctypes.CDLL(os.environ["PFV_HIP_LIB"])
n_rows = 1 << 18
rows = np.zeros((n_rows, 16), np.uint64)
ss.enc.encode_iframe_dev(ss.frame_ptr(0), ss.coef)
ss.enc.encode_pframe_dev(ss.frame_ptr(1), ss.mv, ss.has, ss.coef)
ss.enc.encode_pframe_dev(ss.frame_ptr(2), ss.mv, ss.has, ss.coef)     # the launch that is analysed (second p-frame: warm)
ctx.sync()
assert lib.pfv_debug_kprof(rows.ctypes.data_as(ctypes.c_void_p), n_rows) == 0
r = rows.astype(np.int64)
live = (r[:, 0] > 0) & (r[:, 11] > 0)
r = r[live]
print(f"wavefronts with stamps: {len(r)}")
hw, xcc = r[:, 12], r[:, 13] & 15
simd_key = (xcc << 20) | (((hw >> 8) & 0xff) << 4) | ((hw >> 4) & 3)      # XCC | se, sh, cu | simd
t0, t1, t2, t7, t8, t11 = (r[:, k] for k in (0, 1, 2, 7, 8, 11))
t1 = np.maximum(t1, t0); t2 = np.maximum(t2, t1); t7 = np.maximum(t7, t2); t8 = np.maximum(t8, t7); t11 = np.maximum(t11, t8)
life = (t11 - t0).astype(float)
print(f"wavefront lifetime: mean {life.mean():.0f} cycles; issue loads {100 * ((t1 - t0) / life).mean():.1f} %, fill wait {100 * ((t2 - t1) / life).mean():.1f} %, "
      f"search {100 * ((t7 - t2) / life).mean():.1f} %, release barrier {100 * ((t8 - t7) / life).mean():.1f} %, transform {100 * ((t11 - t8) / life).mean():.1f} %")
keys = np.unique(simd_key)
print(f"SIMDs seen: {len(keys)}")
MAXW = 12
hist = np.zeros((MAXW, MAXW))            # [resident][runnable] -> cycles
why = {"all in fill": 0.0, "all in release barrier": 0.0, "fill + release": 0.0}
span_total = 0.0
for k in keys:
    sel = simd_key == k
    a0, a1, a2, a7, a8, a11 = t0[sel], t1[sel], t2[sel], t7[sel], t8[sel], t11[sel]
    # events: (time, d_resident, d_fill, d_release)
    ev_t = np.concatenate([a0, a11, a1, a2, a7, a8])
    n = len(a0)
    z = np.zeros(n, np.int64); o = np.ones(n, np.int64)
    d_res = np.concatenate([o, -o, z, z, z, z])
    d_fill = np.concatenate([z, z, o, -o, z, z])
    d_rel = np.concatenate([z, z, z, z, o, -o])
    order = np.argsort(ev_t, kind="stable")
    ev_t, d_res, d_fill, d_rel = ev_t[order], d_res[order], d_fill[order], d_rel[order]
    res, fil, rel = np.cumsum(d_res), np.cumsum(d_fill), np.cumsum(d_rel)
    dt = np.diff(ev_t).astype(float)
    res, fil, rel = res[:-1], fil[:-1], rel[:-1]
    run = res - fil - rel
    span_total += ev_t[-1] - ev_t[0]
    np.add.at(hist, (np.minimum(res, MAXW - 1), np.minimum(np.maximum(run, 0), MAXW - 1)), dt)
    none = (res > 0) & (run <= 0)
    why["all in fill"] += dt[none & (rel == 0)].sum()
    why["all in release barrier"] += dt[none & (fil == 0)].sum()
    why["fill + release"] += dt[none & (fil > 0) & (rel > 0)].sum()
tot = hist.sum()
print(f"\\nper-SIMD time, all SIMDs pooled (100 % = {tot / len(keys):.0f} cycles per SIMD between its first wavefront's start and its last one's end)")
print("resident wavefronts:   " + "  ".join(f"{k}: {100 * hist[k].sum() / tot:5.1f} %" for k in range(7)))
print("runnable wavefronts:   " + "  ".join(f"{k}: {100 * hist[:, k].sum() / tot:5.1f} %" for k in range(7)))
print(f"mean resident {sum(k * hist[k].sum() for k in range(MAXW)) / tot:.2f}, mean runnable {sum(k * hist[:, k].sum() for k in range(MAXW)) / tot:.2f}")
print(f"NO wavefront resident: {100 * hist[0].sum() / tot:.1f} %")
print(f"wavefronts resident but NONE runnable: {100 * (hist[1:, 0].sum()) / tot:.1f} %  (" + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in why.items()) + ")")
print(f"exactly ONE runnable (a lone wavefront issues a VALU instruction every ~5 cycles, not 4: 20 % of those slots stay empty): {100 * hist[:, 1].sum() / tot:.1f} %")
idle = hist[:, 0].sum() / tot + 0.2 * hist[:, 1].sum() / tot
print(f"=> VALU issue slots these waits alone leave empty: {100 * idle:.1f} %   (PMC: 1 - SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = the measured idle share)")
ss.close(); ctx.close()
