# A/B of the working tree against other builds of the library on the SAME GPU box, interleaved (boxes differ by +-1.5 %):
#   bash tools/ab_prev.sh [name ...]     names = pretty-fast-video_amd/libpfv_<name>.so made by tools/build_prev.sh (default: prev)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
NAMES=${@:-prev}
for i in 1 2 3; do
  for v in default $NAMES; do
    if [ $v = default ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=$GRAFT_REPO_ROOT/pretty-fast-video_amd/libpfv_$v.so PFV_HIP_LIB_OLDER=1; fi
    python bench.py --no-cpu-baseline --no-entropy --no-extra --steps 10 $AB_BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s' % '$v', round(d['value']/1e6,1), {k: round(v['avg_launch_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
