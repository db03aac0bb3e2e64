#!/bin/bash
# Build the kernels of an earlier commit into pretty-fast-video_amd/libpfv_prev.so (git-ignored, ships with gpurun) for a same-box
# A/B with tools/ab_prev.sh.  usage (in the build container): bash tools/build_prev.sh <commit> [name]
set -e
C=${1:-HEAD}; NAME=${2:-prev}
R=$(cd $(dirname $0)/.. && pwd)
W=/tmp/pfv_prev_$$
rm -rf $W && mkdir -p $W && git -C $R archive $C pretty-fast-video_amd/csrc include | tar -x -C $W
(cd $W/pretty-fast-video_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DPFV_BUILD_ID="\"$NAME\"" -o $R/pretty-fast-video_amd/libpfv_$NAME.so pfv_capi.hip)
rm -rf $W; ls -la $R/pretty-fast-video_amd/libpfv_$NAME.so
