#!/bin/bash
# timeline of pfv_gop_encoder fed from HBM (config 4): kernels and copies of the LAST frames-in-HBM encode pass, in time order
#   usage: gpurun -- 'bash tools/gpu_enc_hbm_timeline.sh [gops_per_batch]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/enc_hbm_timeline
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
cd /tmp && export TMPDIR=/tmp
for G in ${1:-10} ${2:-20}; do
rm -rf /tmp/tl
PFV_E2E_STOP_AFTER_ENCODE=1 PFV_E2E_HBM_GOPS=$G timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d /tmp/tl -o tl -- /tmp/e2e_native 3840 2160 300 15 5 10 20 15 > $OUT/run_$G.json 2> $OUT/run.err
tail -2 $OUT/run.err
python $R/tools/trace_timeline.py /tmp/tl 30 0 > $OUT/timeline_$G.txt
tail -1 $OUT/timeline_$G.txt
done
