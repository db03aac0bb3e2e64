#!/bin/bash
# Build the -DPFV_ENT_PROFILE variant on the GPU box and run tools/ent_profile.py.  usage: bash tools/gpu_entprof.sh <tag> [extra -D flags]
TAG=${1:-entprof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R/pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DPFV_ENT_PROFILE "$@" -o /tmp/libpfv_prof.so pfv_capi.hip 2>$OUT/build.err || { tail -5 $OUT/build.err; exit 1; }
cd $R && PFV_HIP_LIB=/tmp/libpfv_prof.so timeout 600 python tools/ent_profile.py 2>&1 | tee $OUT/ent_profile.txt
