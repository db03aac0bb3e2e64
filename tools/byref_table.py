#!/usr/bin/env python3
"""Table of tools/gpu_byref_quality.sh's JSON lines (one run of tools/e2e_native.cpp under PFV_E2E_STOP_AFTER_ENCODE per quality)."""
import json
import sys

print("quality  stream MB  copied G/s  by-ref G/s  by-ref ms  kernel-wait ms  download-wait ms  batches made again")
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    hs = d["by_reference_host_seconds"]
    print(f'{d["quality"]:7d}  {d["stream_bytes"] / 1e6:9.1f}  {d["encode_value_frames_in_hbm"] / 1e9:10.3f}  {d["by_reference"] / 1e9:10.3f}  '
          f'{d["by_reference_s"] * 1e3:9.2f}  {hs["kernel_wait_s"] * 1e3:14.2f}  {hs["payload_download_s"] * 1e3:16.2f}  {hs["batches_redone"]:18.0f}')
