#!/bin/bash
# Builds and runs every micro-benchmark under tools/ubench on the GPU box; outputs -> gpurun_out/<tag>/ubench_*.txt
# (copied to profiles/ by hand).  usage: bash tools/gpu_ubench.sh <tag>
TAG=${1:-ubench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for f in $R/tools/ubench/*.hip; do
  n=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $f -o /tmp/ub_$n 2> $OUT/ubench_$n.build.log || { echo "build failed: $n"; continue; }
  timeout 300 /tmp/ub_$n > $OUT/ubench_$n.txt 2>&1
  echo "== $n"; head -80 $OUT/ubench_$n.txt
done
