#!/usr/bin/env python3
"""Where one frame of the stream Decoder's device half goes (1080p, one stream, pinned host buffers): sparse / dense
coefficient upload + decode kernel, and the retframe download.  Run on the GPU box: python tools/dec_steps.py"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

g.build_hip()
pkg = g.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
W, H, Q = 1920, 1080, 5
st = pkg.SyntheticStream(W, H)
res = {}
with pkg.Context(0) as ctx:
    lib = ctx._lib
    enc = pkg.EncoderSession(ctx, W, H, Q, 1)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), 1)
    enc.encode_iframe(st.frame(0))
    mv, has, coef = enc.encode_pframe(st.frame(1))
    dec.decode_iframe(enc.encode_iframe(st.frame(0)))
    nb = enc.total_blocks

    def pinned(a):
        p = ctx.host_array(a.nbytes).view(a.dtype).reshape(a.shape)
        p[...] = a
        return p
    flat = coef.reshape(-1)
    idx = np.flatnonzero(flat).astype(np.uint32)
    val = flat[idx]
    p_coef, p_mv, p_has, p_idx, p_val = pinned(coef), pinned(mv), pinned(has), pinned(idx), pinned(val)
    out = ctx.host_array(enc.frame_bytes)
    qidx = np.array([2, 3, 3], np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def timed(fn, n=100):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3
    res["nonzero_fraction"] = float(idx.size / flat.size)
    res["pframe_dense_upload_decode_ms"] = timed(lambda: lib.pfv_dec_pframe(dec.handle, P(p_mv), P(p_has), P(p_coef), P(qidx)))
    res["pframe_sparse_upload_decode_ms"] = timed(lambda: lib.pfv_dec_pframe_sparse(dec.handle, P(p_mv), P(p_has), P(p_idx), P(p_val), idx.size, P(qidx)))
    res["get_frame_download_ms"] = timed(lambda: lib.pfv_dec_get_frame(dec.handle, P(out)))
    res["ctx_sync_only_ms"] = timed(lambda: lib.pfv_ctx_sync(ctx.handle))
print(json.dumps(res, indent=1))
