#!/bin/bash
# Entropy-stage iteration: entropy parity tests, the bench's encode_to_payload figure, and a kernel trace of it.
# usage: bash tools/gpu_ent.sh <tag>
TAG=${1:-ent}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log; fi
echo "== bench"; timeout 600 python bench.py --no-extra --no-cpu-baseline 2>$OUT/bench.err > $OUT/bench.json; tail -2 $OUT/bench.err
python -c "
import json
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()})
e=r.get('encode_to_payload',{}); print('entropy', e.get('value'), e.get('two_stream_value'))"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o ent -- python $R/bench.py --no-extra --no-cpu-baseline --no-two-stream --steps 3 --warmup 1 > $OUT/prof.log 2>&1 )
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -14 $OUT/kernel_stats.csv | cut -c1-160
