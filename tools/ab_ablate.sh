# k_enc_pframe ablations on one box (results invalid by construction; timing only)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for v in NOSTORE STORE_SMALL NOHDR SEARCH1 NOXFORM NOWIN; do (cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DPFV_ABL_$v -o /tmp/lib_$v.so pfv_capi.hip) 2>/dev/null; done
for v in full NOSTORE STORE_SMALL NOHDR SEARCH1 NOXFORM NOWIN full; do
  if [ $v = full ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=/tmp/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-entropy --no-verify --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['kernels']['k_enc_pframe']['avg_launch_ms']*1000,1))"
done
