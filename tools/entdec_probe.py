#!/usr/bin/env python3
"""config 4 (3840x2160, GOP-15, quality 5) .pfv stream through pfv_gop_decoder with the packet payloads read by the host pool and by the
device stage (PFV_OPT_ENTROPY_DECODE): frames must be the same bytes; seconds by what the host waited for.
    python tools/entdec_probe.py [frames] [width height]"""
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as g   # noqa: E402

g.build_hip()
pkg = g.load_package()
__import__("libswitch").apply_from_env(pkg)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
Q, GOP = 5, 15
n_mb = int(pkg._lib.load().pfv_total_blocks(W, H))
st = pkg.SyntheticStream(W, H)
with pkg.Context(0) as ctx:
    buf = io.BytesIO()
    enc = pkg.GopEncoder(buf, W, H, 30, Q, ctx, max_gops=10, max_gop_frames=GOP)
    fr = [pkg.VideoFrame.from_packed(W, H, st.frame(t)) for t in range(GOP)]
    for t in range(N):
        (enc.encode_iframe if t % GOP == 0 else enc.encode_pframe)(fr[t % GOP])
    enc.finish()
    enc.close()
    data = buf.getvalue()
    print("stream", len(data), "bytes,", N, "frames", flush=True)
    if os.environ.get("PFV_PROBE_COUNT_VALUES") == "1":   # what the entropy stage has to deliver: the non-zero coefficients of every packet (host parser)
        import ctypes
        L = pkg._lib.load()
        pos = 20 + 128 * (data[18] | data[19] << 8)
        raw = np.frombuffer(data, np.uint8)
        idx = np.empty(n_mb * 256, np.uint32); val = np.empty(n_mb * 256, np.int16)
        mv = np.empty(n_mb * 2, np.int8); has = np.empty(n_mb, np.uint8); q = np.zeros(3, np.uint8)
        per_type = {1: [0, 0, 0], 2: [0, 0, 0]}
        while pos + 5 <= len(data) and data[pos] != 0:
            typ, plen = data[pos], int.from_bytes(data[pos + 1:pos + 5], "little")
            n = ctypes.c_size_t(0)
            rc = L.pfv_parse_payload_sparse(int(typ == 2), raw[pos + 5:pos + 5 + plen].ctypes.data_as(ctypes.c_void_p), plen, n_mb, 4, mv.ctypes.data_as(ctypes.c_void_p),
                                            has.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p), idx.size, ctypes.byref(n), q.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0, rc
            t = per_type[typ]; t[0] += 1; t[1] += n.value; t[2] += plen
            pos += 5 + plen
        for typ, (k, nv, nb) in per_type.items():
            if k:
                print(f"packets type {typ}: {k}, values per packet {nv / k:.0f} (= {2 * nv / k / 1e6:.3f} MB of int16), payload bytes per packet {nb / k:.0f}", flush=True)
    digests = {}
    for mode in os.environ.get("PFV_PROBE_MODES", "host,device,device,host,device,device->HBM,device->HBM").split(","):
        import hashlib
        h = hashlib.sha256()
        cnt = [0]

        to_hbm = mode.endswith("HBM")
        tmp = np.empty(W * H * 3 // 2, np.uint8)

        def onvideo(y, u, v):
            cnt[0] += 1
            if cnt[0] % 37 == 1:
                if to_hbm:
                    ctx.download(tmp, y)
                    h.update(tmp)
                else:
                    h.update(y); h.update(u); h.update(v)
        d = pkg.GopDecoder(data, ctx, max_gops=20, max_gop_frames=GOP, threads=15, raw=True, entropy=mode.split("-")[0], output="device" if to_hbm else "host")
        t0 = time.perf_counter()
        while d.advance_frame(onvideo):
            pass
        el = time.perf_counter() - t0
        s = d.stats()
        d.close()
        assert cnt[0] == N
        digests.setdefault(mode, set()).add(h.hexdigest())
        print(mode, f"{N * n_mb / el / 1e6:7.1f} M macroblocks/s  {el * 1e3:7.1f} ms", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()}), flush=True)
    assert len(set.union(*digests.values())) == 1, digests
    print("frames identical across modes")
