#!/bin/bash
# The product's two-step build of libpfv_hip.so (csrc/pfv_penc.hip on its own with the max-ILP scheduling strategy, then the rest + link)
# into <out.so>.  usage: bash tools/build_lib.sh <out.so> [flags...]
#   plain flags go to BOTH compilations; --penc=<flag> only to the p-frame encoder's (repeatable); --penc-sched=<flags> replaces its
#   scheduling flags ("--penc-sched=" = the default strategy); -DPFV_NO_PENC_SCHED is shorthand for that.
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCHED="-mllvm -amdgpu-sched-strategy=max-ilp"
BOTH=(); PENC=()
for a in "$@"; do
  case "$a" in
    --penc-sched=*) SCHED="${a#--penc-sched=}";;
    --penc=*) PENC+=("${a#--penc=}");;
    -DPFV_NO_PENC_SCHED) SCHED="";;
    *) BOTH+=("$a");;
  esac
done
cd $R/pretty-fast-video_amd/csrc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPFV_SPLIT_PENC "${BOTH[@]}" $SCHED "${PENC[@]}" -c -o $OUT.penc.o pfv_penc.hip || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPFV_SPLIT_PENC "${BOTH[@]}" -shared -o $OUT $OUT.penc.o pfv_capi.hip || exit 1
rm -f $OUT.penc.o
