#!/bin/bash
# Build the -DPFV_KPROF=2 variant (one stamp row per wavefront) on the GPU box and run tools/kprof_simd.py.  usage: bash tools/gpu_kprof2.sh <tag> [extra -D flags]
TAG=${1:-kprof2}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash $R/tools/build_lib.sh /tmp/libpfv_kprof2.so -DPFV_KPROF=2 "$@" 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
cd $R && PFV_HIP_LIB=/tmp/libpfv_kprof2.so timeout 600 python tools/kprof_simd.py 2>&1 | tee $OUT/kprof_simd.txt
