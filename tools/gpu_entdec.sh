#!/bin/bash
# round 4: the decoder's entropy stage on the device (k_entd_*): parity tests of the GOP objects, then config 4 end to end with the
# payloads read on the host / on the device.   usage: gpurun -- 'bash tools/gpu_entdec.sh'
set -u
O=gpurun_out/entdec; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gop or Gop" > $O/pytest_gop.log 2>&1; tail -3 $O/pytest_gop.log
timeout 900 python tools/entdec_probe.py > $O/probe.log 2>&1; tail -40 $O/probe.log
