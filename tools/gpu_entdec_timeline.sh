#!/bin/bash
# timeline of the GOP decoder's device-entropy path, frames left in HBM (config 4): kernels and copies of the last decode pass, in time order
#   usage: gpurun -- 'bash tools/gpu_entdec_timeline.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/entdec_timeline
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
PFV_E2E_ONLY=payloads_read_on_device_frames_left_in_hbm timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d /tmp/tl -o tl -- /tmp/e2e_native 3840 2160 300 15 5 10 20 15 > $OUT/run.json 2> $OUT/run.err
tail -2 $OUT/run.err; cat $OUT/run.json | cut -c1-600
python $R/tools/trace_timeline.py /tmp/tl ${1:-25} ${2:-0} > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
