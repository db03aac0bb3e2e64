#!/bin/bash
# timeline of pfv_gop_encoder reading its frames BY REFERENCE (config 4's clip, 20 GOPs per launch): kernels and copies of the last passes in
# time order, and the host-side log of the object (PFV_GOP_TRACE) beside it.   gpurun -- 'bash tools/gpu_byref_timeline.sh [quality [frames]]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; Q=${1:-5}; N=${2:-300}
OUT=$R/gpurun_out/byref_timeline; mkdir -p $OUT
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
PFV_GOP_TRACE=1 PFV_E2E_STOP_AFTER_ENCODE=1 PFV_E2E_HBM_GOPS=20 timeout 300 /tmp/e2e_native 3840 2160 $N 15 $Q 10 20 15 > $OUT/hostlog_run.json 2> $OUT/hostlog.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
PFV_E2E_STOP_AFTER_ENCODE=1 PFV_E2E_HBM_GOPS=20 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d /tmp/tl -o tl -- /tmp/e2e_native 3840 2160 $N 15 $Q 10 20 15 > $OUT/run.json 2> $OUT/run.err
tail -2 $OUT/run.err
python $R/tools/trace_timeline.py /tmp/tl $((N / 6)) 0 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt; cat $OUT/run.json
