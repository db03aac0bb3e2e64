# k_enc_pframe experiments on one box (ablated builds give invalid results by construction; timing only)
# usage: bash tools/ab_ablate.sh "NAME:-DFLAG ..." ...
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
names="full"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $flags -o /tmp/lib_$name.so pfv_capi.hip) 2>/dev/null && names="$names $name"
done
for rep in 1 2; do
  for v in $names; do
    if [ $v = full ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=/tmp/lib_$v.so; fi
    python bench.py --no-cpu-baseline --no-entropy --no-verify --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['kernels']['k_enc_pframe']['avg_launch_ms']*1000,1))"
  done
done
