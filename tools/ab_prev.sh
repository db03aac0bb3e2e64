# A/B of the working tree against a previous commit's kernels on the same GPU box: bash tools/ab_prev.sh <commit>
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for i in 1 2 3; do
  for v in default prev; do
    if [ $v = default ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=$GRAFT_REPO_ROOT/pretty-fast-video_amd/libpfv_prev.so; fi
    python bench.py --no-cpu-baseline --no-entropy --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), {k: round(v['avg_launch_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
