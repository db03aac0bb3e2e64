#!/usr/bin/env python3
"""Single 1080p stream (the reference's calling pattern, one Encoder per stream): GOP-15 encode+decode at kernel scope with one launch
per frame operation, no event calls; run under `rocprofv3 --kernel-trace --stats` to split the GOP time into kernel execution and gaps.
    python tools/single_stream_probe.py [streams=1] [reps=20]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as g
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g.build_hip()
pkg = g.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
ctx = pkg.Context(0)
if os.environ.get("PFV_PROBE_LANES"):      # force a lane mapping (pfv_kernels.hip, "Lane mappings")
    ctx.set_option(pkg._lib.PFV_OPT_LANE_MAPPING, {"8": pkg._lib.PFV_LANES_PER_MB_8, "16": pkg._lib.PFV_LANES_PER_MB_16}[os.environ["PFV_PROBE_LANES"]])
ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, [pkg.synth.SEED + 17 * k for k in range(S)], bench.GOP)
rate = ss.wall(reps)
ss.verify()
print(json.dumps({"streams": S, "macroblocks_per_s": rate, "us_per_gop": bench.GOP * S * ss.n_mb / rate * 1e6}))
ss.close(); ctx.close()
