#!/bin/bash
# Build the -DPFV_KPROF variant on the GPU box and run tools/kprof.py.  usage: bash tools/gpu_kprof.sh <tag> [extra -D flags]
TAG=${1:-kprof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash $R/tools/build_lib.sh /tmp/libpfv_kprof.so -DPFV_KPROF "$@" 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
cd $R && PFV_HIP_LIB=/tmp/libpfv_kprof.so timeout 600 python tools/kprof.py 2>&1 | tee $OUT/kprof.txt
