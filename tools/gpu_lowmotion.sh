#!/bin/bash
# parity of the skip-aware transform + the low-motion side measurement.  usage: bash tools/gpu_lowmotion.sh <tag>
TAG=${1:-lm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -x -q -k "sparse_coded or low_motion or synth or pframe_plane or golden or session_small or benched" 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-two-stream --no-entropy 2>$OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print("value %.4g  frac %.4f" % (r["value"], r["roofline"]["frac"]), {k: round(v["avg_launch_ms"] * 1e3, 1) for k, v in r["kernels"].items()})
print(json.dumps(r["extra"]["low_motion"], indent=1))
print({k: v for k, v in r["extra"]["single_stream"].items()})
PY
