import json, sys, time
sys.argv = ["bench.py"]
sys.path.insert(0, "/root/repo")
import bench
import __graft_entry__ as graft
pkg = graft.load_package()
__import__("sys").path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
__import__("libswitch").apply_from_env(pkg)      # PFV_HIP_LIB=<variant build> (A/B scripts); the product loader itself has no override
ctx = pkg.Context(0); dctx = pkg.Context(0)
for (W, H, NF) in ((1920, 1080, 15), (3840, 2160, 60)):
    ss = bench.StreamSet(pkg, ctx, W, H, 5, [pkg.synth.SEED], NF, dec_ctx=dctx)
    ss.wall_pipelined(2)
    # enqueue-only time of plain single-stream passes
    ss.sync(); t0 = time.perf_counter()
    for _ in range(6): ss.step()
    t1 = time.perf_counter(); ss.sync(); t2 = time.perf_counter()
    print(W, H, "single stream: enqueue %.1f us per launch, total %.1f us per launch" % ((t1 - t0) / (6 * NF * 2) * 1e6, (t2 - t0) / (6 * NF * 2) * 1e6))
    ss.close()
