#!/bin/bash
# per-kernel durations of ONE 1080p stream (rocprofv3 kernel trace) under both lane mappings.  usage: bash tools/gpu_single_probe.sh <tag>
TAG=${1:-sprobe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for lanes in 8 16; do
  PFV_PROBE_LANES=$lanes timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/p$lanes -o p -- python $R/tools/single_stream_probe.py > $OUT/probe$lanes.log 2>&1
  f=$(find $OUT/p$lanes -name "*kernel_stats.csv" | head -1)
  echo "== lanes per macroblock: $lanes"; [ -n "$f" ] && cut -d, -f1-8 "$f" | head -8; tail -2 $OUT/probe$lanes.log
  [ -n "$f" ] && cp "$f" $OUT/single_stream_lanes${lanes}_kernel_stats.csv
  rm -rf $OUT/p$lanes
done
