#!/bin/bash
# The frames-in-HBM GOP encoder with the frames taken BY REFERENCE, config 4's clip (4K x 300, GOP-15, 20 GOPs per launch), over the quantiser
# qualities: the object hands .pfv BYTES to the host, so its floor is the larger of the kernel time and the time those bytes need on the link.
# Per quality: stream bytes, macroblocks/s copied / by reference, the by-reference pass, the host's waits for the kernels and for the last
# download, batches made again frame by frame (the arena was outgrown).
#   gpurun --timeout 900 -- 'bash tools/gpu_byref_quality.sh gpurun_out/byref'
O=${1:-gpurun_out/byref}; mkdir -p $O; R=$(pwd)
g++ -O2 -std=c++17 -I include tools/e2e_native.cpp -L pretty-fast-video_amd -lpfv_hip -Wl,-rpath,$R/pretty-fast-video_amd -o /tmp/e2e_native || exit 1
: > $O/byref_by_quality.jsonl
for q in ${QUALITIES:-0 2 4 5 6 8 10}; do
  PFV_E2E_STOP_AFTER_ENCODE=1 timeout 300 /tmp/e2e_native 3840 2160 300 15 $q 10 20 15 >> $O/byref_by_quality.jsonl 2>> $O/byref.err || echo "quality $q: rc $?" >> $O/byref.err
done
python3 tools/byref_table.py $O/byref_by_quality.jsonl | tee $O/byref_by_quality.txt
