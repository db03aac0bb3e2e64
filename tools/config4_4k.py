#!/usr/bin/env python3
"""BASELINE.json config #4: 3840x2160 synthetic GOP-15 stream, FULL encode -> .pfv bytes -> decode on one MI355X,
reported at the three scopes of SURVEY.md section 8d:
   (i)   kernels only, frames / coefficients resident in HBM (device-pointer session API, 1 stream),
   (ii)  + PCIe (host-buffer session API),
   (iii) end to end through Encoder / Decoder incl. host RLE + Huffman + bit packing (single host thread).
Every decoded frame is checked against the encoder's closed-loop reconstruction.
    python tools/config4_4k.py [--frames 300]"""
import argparse
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--quality", type=int, default=5)
args = ap.parse_args()

import torch   # device memory plumbing only

g.build_hip()
pkg = g.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
W, H, Q, GOP, N = 3840, 2160, args.quality, 15, args.frames
st = pkg.SyntheticStream(W, H)
t0 = time.perf_counter()
frames = [st.frame(t) for t in range(N)]
t_gen = time.perf_counter() - t0
nmb = 48720
res = {"config": f"{W}x{H}, {N} frames, GOP-{GOP}, quality {Q}", "macroblocks": N * nmb, "synthetic_generation_s": round(t_gen, 1)}

with pkg.Context(0) as ctx:
    dev = torch.device("cuda", 0)
    # ---- (i) kernels only
    d_frames = torch.from_numpy(np.stack(frames[:GOP * 2])).to(dev)      # 2 GOPs resident; the GOP pattern repeats
    enc = pkg.EncoderSession(ctx, W, H, Q, 1)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), 1)
    coef = torch.empty((nmb, 256), dtype=torch.int16, device=dev)
    mv = torch.empty((nmb, 2), dtype=torch.int8, device=dev)
    has = torch.empty((nmb,), dtype=torch.uint8, device=dev)
    out = torch.empty((d_frames.shape[1],), dtype=torch.uint8, device=dev)
    dec.set_output_dev(out.data_ptr())
    torch.cuda.synchronize()
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        for t in range(N):
            f = d_frames[t % (GOP * 2)].data_ptr()
            if t % GOP == 0:
                enc.encode_iframe_dev(f, coef.data_ptr()); dec.decode_iframe_dev(coef.data_ptr())
            else:
                enc.encode_pframe_dev(f, mv.data_ptr(), has.data_ptr(), coef.data_ptr())
                dec.decode_pframe_dev(mv.data_ptr(), has.data_ptr(), coef.data_ptr())
        ctx.sync()
        el = time.perf_counter() - t0
    assert np.array_equal(enc.prev_frame(), dec.framebuffer())
    res["kernel_only_mb_per_s"] = N * nmb / el
    res["kernel_only_note"] = "single 4K stream, one launch per frame operation (48 720 macroblocks per launch: launch-latency bound)"
    enc.close(); dec.close()
    # ---- (ii) + PCIe
    enc = pkg.EncoderSession(ctx, W, H, Q, 1)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), 1)
    M = min(N, 45)
    t0 = time.perf_counter()
    for t in range(M):
        if t % GOP == 0:
            dec.decode_iframe(enc.encode_iframe(frames[t]))
        else:
            dec.decode_pframe(*enc.encode_pframe(frames[t]))
        dec.get_frame()
    el = time.perf_counter() - t0
    res["pcie_inclusive_mb_per_s"] = M * nmb / el
    enc.close(); dec.close()
    # ---- (iii) end to end with host entropy + container
    buf = io.BytesIO()
    e = pkg.Encoder(buf, W, H, 30, Q, ctx)
    hot = pkg.EncoderSession(ctx, W, H, Q, 1)        # second encoder only to obtain the reconstruction to compare with
    t_enc = 0.0
    recon = []
    for t in range(N):
        vf = pkg.VideoFrame.from_packed(W, H, frames[t])
        t0 = time.perf_counter()
        (e.encode_iframe if t % GOP == 0 else e.encode_pframe)(vf)
        t_enc += time.perf_counter() - t0
        if t < 30:                                   # keep the check affordable: first two GOPs
            (hot.encode_iframe if t % GOP == 0 else hot.encode_pframe)(frames[t])
            pf = pkg.VideoFrame.from_packed(W, H, hot.prev_frame()[0], padded=True)
            recon.append(np.concatenate([pf.plane_y.image()[:H, :W].reshape(-1), pf.plane_u.image()[:H // 2, :W // 2].reshape(-1),
                                         pf.plane_v.image()[:H // 2, :W // 2].reshape(-1)]))
    e.finish(); e.close(); hot.close()
    data = buf.getvalue()
    d = pkg.Decoder(data, ctx)
    got = []
    n = [0]

    def onvideo(fr):
        if n[0] < 30:
            got.append(fr.packed())
        n[0] += 1
    t0 = time.perf_counter()
    while d.advance_frame(onvideo):
        pass
    t_dec = time.perf_counter() - t0
    d.close()
    assert n[0] == N and all(np.array_equal(a, b) for a, b in zip(got, recon)), "decoded stream != encoder reconstruction"
    res.update({"stream_bytes": len(data), "bits_per_pixel": round(len(data) * 8 / (N * W * H), 3),
                "end_to_end_encode_mb_per_s": N * nmb / t_enc, "end_to_end_decode_mb_per_s": N * nmb / t_dec,
                "end_to_end_encdec_mb_per_s": N * nmb / (t_enc + t_dec)})
print(json.dumps(res, indent=1))
