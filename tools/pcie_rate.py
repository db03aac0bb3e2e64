#!/usr/bin/env python3
"""PCIe-inclusive and end-to-end (host entropy) rates of the host-buffer entry points -- NEVER the bench `value`,
recorded in profiles/README.md per DESIGN.md section 5.  Run on the GPU box: python tools/pcie_rate.py"""
import io
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
W, H, Q, GOP, S = 1920, 1080, 5, 15, 8
st = pkg.SyntheticStream(W, H)
frames1 = [st.frame(t) for t in range(GOP)]
res = {}
with pkg.Context(0) as ctx:
    # (ii) device kernels + PCIe: host buffers in, coefficients / headers out, S streams per call
    enc = pkg.EncoderSession(ctx, W, H, Q, S)
    dec = pkg.DecoderSession(ctx, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), S)
    batch = [np.tile(f, (S, 1)) for f in frames1]
    for rep in range(3):
        t0 = time.perf_counter()
        for t, f in enumerate(batch):
            if t == 0:
                dec.decode_iframe(enc.encode_iframe(f))
            else:
                dec.decode_pframe(*enc.encode_pframe(f))
            dec.get_frame()
        el = time.perf_counter() - t0
    res["pcie_inclusive_mb_per_s"] = GOP * S * enc.total_blocks / el
    res["pcie_inclusive_note"] = f"{S} streams, pageable host buffers, synchronous host-pointer entry points, encode+decode+retframe download"
    enc.close(); dec.close()
    # (iii) end to end through the stream objects: one 1080p stream, 4 GOPs, Encoder -> .pfv bytes -> Decoder
    import ctypes
    lib = ctx._lib
    NG = 4
    vf = [pkg.VideoFrame.from_packed(W, H, f) for f in frames1]
    nmb = NG * GOP * 12240
    data = None
    for mode, dev in (("host_entropy", False), ("device_entropy", True)):
        best = 0.0
        for rep in range(2):
            buf = io.BytesIO()
            e = pkg.Encoder(buf, W, H, 30, Q, ctx, device_entropy=dev)
            t0 = time.perf_counter()
            for g_ in range(NG):
                for t, f in enumerate(vf):
                    (e.encode_iframe if t == 0 else e.encode_pframe)(f)
            e.finish()
            best = max(best, nmb / (time.perf_counter() - t0))
            e.close()
            assert data is None or data == buf.getvalue()      # both entropy paths write the same bytes
            data = buf.getvalue()
        res[f"end_to_end_encode_{mode}_mb_per_s"] = best
    arr = np.frombuffer(data, np.uint8)
    for name, la in (("inline_parse", 0), ("lookahead_4", 4), ("lookahead_8", 8)):
        best = 0.0
        for rep in range(2):
            d = pkg.Decoder(data, ctx, lookahead=la)
            n = 0
            t0 = time.perf_counter()
            while lib.pfv_decoder_advance_frame(d.handle, None, None) == 1:      # C entry point, no Python callback
                n += 1
            best = max(best, nmb / (time.perf_counter() - t0))
            d.close()
            assert n == NG * GOP
        res[f"end_to_end_decode_{name}_mb_per_s"] = best
    # (iv) batch encoder: S streams per step, one upload / launch set / download; frames written straight into the
    # page-locked array, writers are in-memory
    for SB in (8, 32):
        bufs = [io.BytesIO() for _ in range(SB)]
        be = pkg.BatchEncoder(bufs, W, H, 30, Q, ctx)
        fr = []
        for f in frames1:                                   # the producer's frames, already in page-locked memory
            a = ctx.host_array(SB * f.size).reshape(SB, f.size)
            a[...] = f
            fr.append(a)
        best = 0.0
        for rep in range(2):
            t0 = time.perf_counter()
            for t in range(GOP):
                (be.encode_iframes if t == 0 else be.encode_pframes)(fr[t])
            best = max(best, GOP * SB * 12240 / (time.perf_counter() - t0))
        be.close()
        res[f"batch_encoder_{SB}_streams_mb_per_s"] = best
        # (v) batch decoder over those streams: per-stream bit parsing on a thread pool, one sparse upload + one launch per step
        for th in (8, 16):
            bd = pkg.BatchDecoder([b.getvalue() for b in bufs], ctx, threads=th)
            t0 = time.perf_counter()
            steps = 0
            while bd.advance_frames() is not False:
                steps += 1
            el = time.perf_counter() - t0
            bd.close()
            res[f"batch_decoder_{SB}_streams_{th}_threads_mb_per_s"] = steps * SB * 12240 / el
    res.update({"stream_bytes": len(data),
                "end_to_end_note": "one 1080p stream, 4 x GOP-15, pinned staging; encode = upload + kernels (+ device entropy | + "
                                   "coefficient download + host entropy) + packet assembly; decode = host bit parser (inline or on "
                                   "look-ahead threads) + upload + kernels + retframe download"})
print(json.dumps(res, indent=1))
