#!/usr/bin/env python3
"""Merge rocprofv3 --kernel-trace --memory-copy-trace CSVs into one time-ordered list of the LAST `window_ms` milliseconds of device activity:
    python tools/trace_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> [window_ms] [min_us]
One line per event: start (us from the window's first event), duration, queue / copy direction, name or bytes.  Measurement tool."""
import csv
import glob
import os
import sys

d = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pfv::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), name))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy").replace("MEMORY_COPY_", ""), "copy"))
ev.sort()
if not ev:
    sys.exit("no events")
t_end = ev[-1][1]
ev = [e for e in ev if e[0] >= t_end - window_ms * 1e6]
t0 = ev[0][0]
for s, e, q, name in ev:
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  {q:16s} {name}")
print(f"window: {(t_end - t0) / 1e6:.3f} ms, {len(ev)} events")
