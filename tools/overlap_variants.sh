cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
mkdir -p gpurun_out/ovl
python tools/overlap_probe.py 2>gpurun_out/ovl/default.err | tail -1
for spec in "w4:-DPFV_PENC_WAVES=4" "pad36:-DPFV_PENC_LDS_PAD=18432" "pad28:-DPFV_PENC_LDS_PAD=10240" "w4pad:-DPFV_PENC_WAVES=4 -DPFV_PENC_LDS_PAD=18432" "w3:-DPFV_PENC_WAVES=3"; do
  name=${spec%%:*}; flags=${spec#*:}
  (cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $flags -o /tmp/lib_$name.so pfv_capi.hip) 2>gpurun_out/ovl/$name.build.err || { echo "$name build failed"; continue; }
  PFV_HIP_LIB=/tmp/lib_$name.so python tools/overlap_probe.py 2>gpurun_out/ovl/$name.err | tail -1
done
