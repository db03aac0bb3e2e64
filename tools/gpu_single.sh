#!/bin/bash
# single-stream / config-4 side measurements of the bench (no headline extras).  usage: bash tools/gpu_single.sh <tag>
TAG=${1:-single}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
python - <<'PY' 2>&1 | tee $OUT/single.txt
import json, sys, time
sys.argv = ["bench.py"]
import bench
import __graft_entry__ as graft
pkg = graft.load_package()
ctx = pkg.Context(0)
print(json.dumps(bench.single_stream_side(pkg, ctx, 5), indent=1))
r = bench.stream_4k_side(pkg, ctx, 5, pkg.synth.SEED, n_frames=60, pcie_frames=2)
print(json.dumps(r["kernel_only"], indent=1))
ctx.close()
PY
