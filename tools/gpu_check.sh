#!/bin/bash
# Standard GPU-box sequence: parity tests, smoke, bench, rocprofv3 kernel stats.
# usage (from the repo root on the GPU box): bash tools/gpu_check.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py "$@" 2>$OUT/bench.err | tee $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stream "$@" > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
