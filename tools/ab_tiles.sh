# k_enc_pframe with 1 / 2 / 4 tiles per workgroup on one box
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for t in 2 4; do (cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DPFV_PENC_TILES=$t -o /tmp/lib_t$t.so pfv_capi.hip) 2>/dev/null; done
for i in 1 2; do
  for v in 1 2 4; do
    if [ $v = 1 ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=/tmp/lib_t$v.so; fi
    python bench.py --no-cpu-baseline --no-entropy --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiles/WG $v', round(d['value']/1e6,1), {k: round(v['avg_launch_ms']*1000,1) for k,v in d['kernels'].items()})"
  done
done
