#!/usr/bin/env python3
"""tools/sanitize.sh, ThreadSanitizer part: builds the emulator library and tests/cpp/roundtrip.cpp (the C++ mirror's round trip: Encoder,
Decoder with its look-ahead threads, a damaged stream, GopEncoder / GopDecoder with its parse pool, BatchEncoder / BatchDecoder with their
pools) with the flags in PFV_EMU_DEFS and writes the program's input clip.  The program then runs NATIVELY -- no python, no preloaded
runtime: gcc 11's libtsan preloaded into python hangs at start-up in this container.  Prints the program's path."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import conftest                    # noqa: E402
import __graft_entry__ as g       # noqa: E402

flags = os.environ.get("PFV_EMU_DEFS", "").split()
out = sys.argv[1]
lib = conftest.build_emulator()
exe = os.path.join(out, "roundtrip_san")
subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "roundtrip.cpp"), "-o", exe, lib,
                "-Wl,-rpath," + os.path.dirname(lib)], check=True)
drv = os.path.join(out, "threads_driver_san")     # every threaded object, intact + damaged streams (one process per stream: tools/sanitize.sh)
subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "threads_driver.cpp"), "-o", drv, lib,
                "-Wl,-rpath," + os.path.dirname(lib)], check=True)
pkg = g.load_package()
w, h, n = 64, 48, 12
st = pkg.SyntheticStream(w, h)
np.concatenate([st.frame(t) for t in range(n)]).tofile(os.path.join(out, "in.yuv"))
print(exe)
