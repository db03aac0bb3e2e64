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
    # (iii) end to end with host entropy + container: one stream through Encoder / Decoder
    buf = io.BytesIO()
    e = pkg.Encoder(buf, W, H, 30, Q, ctx)
    vf = [pkg.VideoFrame.from_packed(W, H, f) for f in frames1]
    t0 = time.perf_counter()
    for t, f in enumerate(vf):
        (e.encode_iframe if t == 0 else e.encode_pframe)(f)
    e.finish()
    t_enc = time.perf_counter() - t0
    e.close()
    data = buf.getvalue()
    d = pkg.Decoder(data, ctx)
    n = [0]
    t0 = time.perf_counter()
    while d.advance_frame(lambda fr: n.__setitem__(0, n[0] + 1)):
        pass
    t_dec = time.perf_counter() - t0
    d.close()
    assert n[0] == GOP
    nmb = GOP * 12240
    res.update({"end_to_end_encode_mb_per_s": nmb / t_enc, "end_to_end_decode_mb_per_s": nmb / t_dec,
                "end_to_end_encdec_mb_per_s": nmb / (t_enc + t_dec), "stream_bytes": len(data),
                "end_to_end_note": "one 1080p stream, GOP-15, host RLE/Huffman/bit-packing (single host thread) + PCIe + kernels"})
print(json.dumps(res, indent=1))
