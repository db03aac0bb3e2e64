#!/bin/bash
# The product's two-step build of libpfv_hip.so (csrc/pfv_penc.hip on its own with the max-ILP scheduling strategy, then the rest + link)
# into <out.so>, with extra flags for BOTH compilations.  usage: bash tools/build_lib.sh <out.so> [flags...]      (PENC_SCHED="" or
# another -mllvm strategy in the environment overrides the p-frame encoder's scheduling flags)
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCHED=${PENC_SCHED--mllvm -amdgpu-sched-strategy=max-ilp}
case " $* " in *" -DPFV_NO_PENC_SCHED "*) SCHED="";; esac      # A/B: the p-frame encoder under the default scheduling strategy
cd $R/pretty-fast-video_amd/csrc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPFV_SPLIT_PENC "$@" $SCHED -c -o $OUT.penc.o pfv_penc.hip || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPFV_SPLIT_PENC "$@" -shared -o $OUT $OUT.penc.o pfv_capi.hip || exit 1
rm -f $OUT.penc.o
