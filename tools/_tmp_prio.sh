set -u
O=gpurun_out/prio; mkdir -p $O
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
for cfg in "0 0" "-1 0" "1 0" "-1 1" "0 1" "0 0"; do
set -- $cfg
for rep in 1 2; do
PFV_GOPD_WIN_PRIO=$1 PFV_GOPD_UP_PRIO=$2 PFV_E2E_HBM_GOPS=20 PFV_E2E_ONLY=payloads_read_on_device_frames_left_in_hbm timeout 600 /tmp/e2e_native 3840 2160 300 15 5 10 20 15 > $O/r.json 2> $O/r.err
python - $O/r.json "$cfg" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
v = r["decode"]["payloads_read_on_device_frames_left_in_hbm"]
print("windows/upload priority %s: %.1f M  %s" % (sys.argv[2], v["decode_value"] / 1e6, {a: round(b * 1e3, 1) for a, b in v["decoder_host_seconds"].items()}))
PY
done
done
