#!/bin/bash
# round 3, first GPU pass: parity suite, bench, kernel stats, the rounding probe.  usage: bash tools/gpu_r3a.sh <tag>
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/cvt_pk_u8_rounding.hip -o /tmp/ub_cvt 2> $OUT/ub_cvt.build.log && /tmp/ub_cvt | tee $OUT/ubench_cvt_pk_u8_rounding.txt
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15) 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-1500; tail -3 $OUT/bench.err
