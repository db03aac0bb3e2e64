#!/bin/bash
# Quick GPU iteration: parity suite + a short bench without the side measurements.
# usage: bash tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-q}; K=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu"; if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8; else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8; fi | tee $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --no-extra --no-cpu-baseline --no-two-stream 2>$OUT/bench.err > $OUT/bench.json; tail -2 $OUT/bench.err; python -c "
import json
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, 'frac', round(r['roofline']['frac'],4), {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()})
print('entropy', r.get('encode_to_payload',{}).get('value'))"
