cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
(cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DPFV_NO_MUL24 -o /tmp/lib_nomul24.so pfv_capi.hip) 2>/dev/null
for i in 1 2 3; do
  for v in default nomul24; do
    if [ $v = default ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=/tmp/lib_nomul24.so; fi
    python bench.py --no-cpu-baseline --no-entropy --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), {k: round(v['avg_launch_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
