#!/bin/bash
# round 4: the decoder's entropy stage on the device (k_entd_*): parity tests of the GOP objects, then config 4 end to end with the
# payloads read on the host / on the device (Python mirror, then the native host program), kernel stats.
#   usage: gpurun -- 'bash tools/gpu_entdec.sh'
set -u
O=gpurun_out/entdec; mkdir -p $O
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
if [ "${1:-}" != "quick" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gop or Gop" > $O/pytest_gop.log 2>&1; tail -3 $O/pytest_gop.log
fi
[ "${1:-}" != "quick" ] && { timeout 900 python tools/entdec_probe.py > $O/probe.log 2>&1; tail -12 $O/probe.log; }
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native
for sb in 128 256; do
  timeout 600 /tmp/e2e_native 3840 2160 300 15 5 10 20 15 $sb > $O/native_$sb.json 2> $O/native.err; tail -3 $O/native.err
  python - $O/native_$sb.json $sb <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("sub_bits", sys.argv[2], "encode %.1f M" % (r["encode_value"] / 1e6))
for k, v in r["decode"].items():
    print("   %-45s %7.1f M  %s left %d (unsettled %d, irregular %d)" % (k, v["decode_value"] / 1e6, {a: round(b * 1e3, 1) for a, b in v["decoder_host_seconds"].items()}, v["packets_left_to_host_parser"], v["left_unsettled"], v["left_irregular"]))
PY
done
cp $O/native_256.json $O/native.json
cd /tmp && export TMPDIR=/tmp
PFV_PROBE_MODES="device->HBM,device->HBM" timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/entdec_prof -o entdec -- python $R/tools/entdec_probe.py 150 > $R/$O/prof.log 2>&1
cd $R
f=$(find /tmp/entdec_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -16 $O/kernel_stats.csv | cut -c1-160
