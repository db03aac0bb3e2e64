import sys, time
sys.argv = ["bench.py"]; sys.path.insert(0, ".")
import bench, __graft_entry__ as graft
pkg = graft.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
ctx = pkg.Context(0); dctx = pkg.Context(0)
from importlib import import_module
shard = import_module("pretty_fast_video_amd.shard")
seeds = [int(r[1]) for r in shard.streams_of_rank(shard.assign_streams(96, 1, pkg.synth.SEED), 0)]
for S in (16, 32, 96):
    ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, seeds[:S], bench.GOP)
    a = ss.wall(6); ss.verify(); ss.close()
    ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, seeds[:S], bench.GOP, dec_ctx=dctx)
    b = ss.wall_pipelined(6); ss.verify(); ss.close()
    print("streams %3d: one stream of launches %.1f M, decoder one GOP behind on a second stream %.1f M macroblocks/s" % (S, a / 1e6, b / 1e6), flush=True)
