#!/usr/bin/env python3
"""Experiment: decode of frame t on a second HIP stream, overlapping the encode of frame t+1 (double-buffered
coefficient / header buffers, event-ordered).  Prints ms per GOP for the one-stream and the two-stream schedule.
Run on the GPU box: [PFV_HIP_LIB=variant.so] python tools/overlap_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g   # noqa: E402

if not os.environ.get("PFV_HIP_LIB"):
    g.build_hip()
pkg = g.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
W, H, S, Q, GOP = 1920, 1080, 32, 5, 15
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
fb = int(pkg._lib.load().pfv_frame_bytes(W, H))
host = np.empty((GOP, 2, fb), np.uint8)
for u in range(2):
    st = pkg.SyntheticStream(W, H, seed=pkg.synth.SEED + u)
    for t in range(GOP):
        host[t, u] = st.frame(t)
frames = torch.from_numpy(host).to(dev)[:, torch.arange(S, device=dev) % 2].contiguous()
ctx_e, ctx_d = pkg.Context(0), pkg.Context(0)
enc = pkg.EncoderSession(ctx_e, W, H, Q, S)
dec = pkg.DecoderSession(ctx_d, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), S)
dec1 = pkg.DecoderSession(ctx_e, W, H, np.stack(pkg.qtables_from_quality(Q)[:4]), S)    # same-stream baseline
n_mb = enc.total_blocks
sets = [(torch.empty((S, n_mb, 256), dtype=torch.int16, device=dev), torch.empty((S, n_mb, 2), dtype=torch.int8, device=dev),
         torch.empty((S, n_mb), dtype=torch.uint8, device=dev)) for _ in range(2)]
out = torch.empty((S, fb), dtype=torch.uint8, device=dev)
dec.set_output_dev(out.data_ptr()); dec1.set_output_dev(out.data_ptr())
se = torch.cuda.ExternalStream(ctx_e.stream, device=dev)
sd = torch.cuda.ExternalStream(ctx_d.stream, device=dev)


def gop_one_stream():
    for t in range(GOP):
        c, m, h = sets[t & 1]
        f = frames[t].data_ptr()
        if t == 0:
            enc.encode_iframe_dev(f, c.data_ptr()); dec1.decode_iframe_dev(c.data_ptr())
        else:
            enc.encode_pframe_dev(f, m.data_ptr(), h.data_ptr(), c.data_ptr())
            dec1.decode_pframe_dev(m.data_ptr(), h.data_ptr(), c.data_ptr())


done = [None, None]      # decode events per buffer set


def gop_two_streams():
    for t in range(GOP):
        c, m, h = sets[t & 1]
        f = frames[t].data_ptr()
        if done[t & 1] is not None:
            se.wait_event(done[t & 1])            # the decoder has finished reading this set
        if t == 0:
            enc.encode_iframe_dev(f, c.data_ptr())
        else:
            enc.encode_pframe_dev(f, m.data_ptr(), h.data_ptr(), c.data_ptr())
        e = torch.cuda.Event(); e.record(se); sd.wait_event(e)
        if t == 0:
            dec.decode_iframe_dev(c.data_ptr())
        else:
            dec.decode_pframe_dev(m.data_ptr(), h.data_ptr(), c.data_ptr())
        d = torch.cuda.Event(); d.record(sd); done[t & 1] = d


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(se)
    for _ in range(reps):
        fn()
    if done[0] is not None:
        se.wait_event(done[0]); se.wait_event(done[1])
    e1.record(se)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


one = timed(gop_one_stream)
two = timed(gop_two_streams)
dec.check(); dec1.check()
ok = bool(np.array_equal(enc.prev_frame(), dec.framebuffer()))
print(json.dumps({"lib": os.environ.get("PFV_HIP_LIB", "default"), "one_stream_ms_per_gop": one, "two_stream_ms_per_gop": two,
                  "one_stream_G_mb_per_s": GOP * S * n_mb / one / 1e6, "two_stream_G_mb_per_s": GOP * S * n_mb / two / 1e6,
                  "decoder_matches_encoder": ok}))
