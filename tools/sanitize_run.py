#!/usr/bin/env python3
"""The host-thread-heavy checks of the CPU suite on the emulator build named by PFV_EMU_DEFS, WITHOUT pytest (its process handling and
ThreadSanitizer's runtime do not get along: the run hangs before the first test) -- tools/sanitize.sh runs this under LD_PRELOAD=libtsan.so.
The checks are the ones of tests/test_emulated_kernels.py: damaged streams through the frame-by-frame decoder with and without look-ahead
threads, the GOP objects (encoder batches, decoder parse pool, device-entropy windows + host-parser fallback), the batch decoder's pool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import conftest            # noqa: E402
import libswitch           # noqa: E402
import __graft_entry__ as g   # noqa: E402
from oracle_bind import Oracle   # noqa: E402
import stream_cases as sc  # noqa: E402

pkg = g.load_package()
lib = conftest.build_emulator()
libswitch.use(pkg, lib)
print("library:", os.path.basename(lib), flush=True)
ctx = pkg.Context(0)
oracle = Oracle()
t0 = time.time()


ONLY = os.environ.get("PFV_SAN_ONLY")     # substring of a step's name: run that step alone


def step(name, fn):
    if ONLY and ONLY not in name:
        return
    t = time.time()
    r = fn()
    print(f"ok  {name}  ({time.time() - t:.1f} s)  {r if isinstance(r, (dict, int)) else ''}", flush=True)


data, _ = sc.encode_clip(pkg, ctx, oracle, 48, 32, 30, 5, n_frames=4, gop=2)
step("corrupted streams (pfv_decoder: inline, 1 and 3 look-ahead threads; host and device entropy)", lambda: sc.check_corrupted_streams(pkg, ctx, oracle, data, n_trials=24, seed=5))
step("look-ahead reset", lambda: sc.check_lookahead_reset(pkg, ctx, data, n_frames=4))
step("stream round trip", lambda: sc.check_stream_roundtrip(pkg, ctx, oracle, 64, 48, 5, n_frames=5, gop=3) and None)
step("GOP objects", lambda: sc.check_gop_objects(pkg, ctx, oracle, 64, 48, 5, "IPPPIPDPPIPPP", shapes=((8, 15), (2, 3)), alternate_modes=True) and None)
data2, _ = sc.encode_pattern(pkg, ctx, oracle, 48, 32, 5, "IPPIPPPIP", lambda buf: pkg.Encoder(buf, 48, 32, 30, 5, ctx), with_oracle=False)
step("GOP decoder, damaged streams (parse pool + device windows + fallback)", lambda: sc.check_gop_decoder_corrupted(pkg, ctx, oracle, data2, n_trials=4, seed=4))
step("GOP decoder, device entropy", lambda: sc.check_gop_device_entropy(pkg, ctx, oracle, 96, 64, pattern="IPPIP"))
step("batch encoder / decoder pools", lambda: (sc.check_batch_encoder(pkg, ctx, oracle, 64, 48, 5, n_streams=3, n_frames=4, gop=2), sc.check_batch_decoder(pkg, ctx, oracle, 64, 48, 5, n_streams=3, n_frames=4, gop=2)) and None)
ctx.close()
print(f"all checks passed in {time.time() - t0:.0f} s", flush=True)
