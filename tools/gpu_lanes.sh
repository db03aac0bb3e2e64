#!/bin/bash
# both lane mappings: full parity suite, then the single-stream / 4K side measurements under auto / 8 / 16 lanes per macroblock
TAG=${1:-lanes}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -x -q) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|real" $OUT/pytest_gpu.log | tail -4
python - <<'PY' 2>&1 | tee $OUT/single.txt
import json, sys, time
sys.argv = ["bench.py"]
import bench
import __graft_entry__ as graft
pkg = graft.load_package()
L = pkg._lib
ctx = pkg.Context(0)
for name, val in (("auto", L.PFV_LANES_AUTO), ("8", L.PFV_LANES_PER_MB_8), ("16", L.PFV_LANES_PER_MB_16)):
    ctx.set_option(L.PFV_OPT_LANE_MAPPING, val)
    out = {}
    for S in (1, 2, 4, 8, 16):
        ss = bench.StreamSet(pkg, ctx, 1920, 1080, 5, [pkg.synth.SEED + 17 * k for k in range(S)], bench.GOP)
        out[S] = round(ss.wall(6) / 1e6, 1)
        ss.verify(); ss.close()
    ss = bench.StreamSet(pkg, ctx, 3840, 2160, 5, [pkg.synth.SEED], 60)
    out["4k"] = round(ss.wall(3) / 1e6, 1)
    ss.verify(); ss.close()
    print("lanes", name, "M MB/s by streams:", out, flush=True)
ctx.close()
PY
