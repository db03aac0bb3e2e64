#!/bin/bash
# experiment: config-4 decode with frames left in HBM under variants of the entropy stage's stream set-up (PFV_DBG_* read by pfv_gop_decoder_create)
R=$PWD
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
run() {
  env "$@" PFV_E2E_ONLY=payloads_read_on_device_frames_left_in_hbm timeout 300 /tmp/e2e_native 3840 2160 300 15 5 10 20 15 > /tmp/o.json 2>/tmp/o.err
  python - "$*" <<'PY'
import json, sys
r = json.load(open("/tmp/o.json"))["decode"]["payloads_read_on_device_frames_left_in_hbm"]
print("%-60s %.1f M" % (sys.argv[1], r["decode_value"] / 1e6), {k: round(v * 1e3, 1) for k, v in r["decoder_host_seconds"].items()}, "left to host", r["packets_left_to_host_parser"])
PY
}
for rep in 1 2; do
for n in 1 2 3 4; do
  run PFV_DBG_ENTD_STREAMS=$n
  run PFV_DBG_ENTD_STREAMS=$n PFV_DBG_ENTD_NOPRIO=1
done
done
