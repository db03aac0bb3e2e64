#!/bin/bash
# round 5: pfv_gop_encoder with the frames in HBM (kernels on the encoder's own stream, frame copies on the caller's): GOP object parity tests,
# then config 4 from the native host program -- the batch widths timed by tools/e2e_native.cpp and the encoder's host seconds.
#   usage: gpurun -- 'bash tools/gpu_enc_hbm.sh [quick]'
set -u
O=gpurun_out/enc_hbm; mkdir -p $O
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
if [ "${1:-}" != "quick" ]; then
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gop or Gop or one_symbol" > $O/pytest_gop.log 2>&1; tail -3 $O/pytest_gop.log
fi
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
for rep in 1 2; do
  PFV_E2E_ONLY=payloads_read_on_device_frames_left_in_hbm timeout 600 /tmp/e2e_native 3840 2160 300 15 5 10 20 15 > $O/native_$rep.json 2> $O/native.err; tail -3 $O/native.err
  python - $O/native_$rep.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("encode %.1f M; frames in HBM %.1f M (%d GOPs per batch) by width %s; first object %.1f M" % (r["encode_value"] / 1e6, r["encode_value_frames_in_hbm"] / 1e6, r["gops_per_batch"]["encoder_frames_in_hbm"],
      {k: round(v / 1e6, 1) for k, v in r["encode_value_frames_in_hbm_by_gops_per_batch"].items()}, r["encode_value_frames_in_hbm_first_object_of_the_process"] / 1e6))
print("   host ms", {k: round(v * 1e3, 2) for k, v in r["encoder_host_seconds_frames_in_hbm"].items()}, "total %.2f" % (r["encode_frames_in_hbm_s"] * 1e3))
PY
done
