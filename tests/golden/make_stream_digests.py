#!/usr/bin/env python3
"""Regression pins of the restated .pfv byte layout (container + entropy coder): SHA-256 of the oracle's stream for a few
small synthetic clips.  They do NOT pin parity with a Rust-built stream (no toolchain, no upstream fixture; DESIGN.md
section 4) -- they make an accidental change of the restatement visible.  Regenerate: python tests/golden/make_stream_digests.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as g   # noqa: E402

CLIPS = [  # (width, height, framerate, quality, n_frames, gop, drop_at)
    (48, 32, 30, 5, 5, 3, 2),
    (64, 48, 24, 0, 3, 15, -1),
    (34, 18, 60, 10, 4, 2, -1),
    (176, 144, 25, 7, 4, 4, 1),
]


def stream_bytes(pkg, oracle, clip):
    from oracle_bind import OracleStreamEncoder
    w, h, fps, q, n, gop, drop = clip
    st = pkg.SyntheticStream(w, h)
    enc = OracleStreamEncoder(oracle, w, h, fps, q)
    for t in range(n):
        if t == drop:
            enc.encode_dropframe()
        elif t % gop == 0:
            enc.encode_iframe(st.frame(t))
        else:
            enc.encode_pframe(st.frame(t))
    enc.finish()
    return enc.bytes()


if __name__ == "__main__":
    from oracle_bind import Oracle
    g.build_oracle()
    pkg, ora = g.load_package(), Oracle()
    out = []
    for c in CLIPS:
        b = stream_bytes(pkg, ora, c)
        out.append({"clip": list(c), "bytes": len(b), "sha256": hashlib.sha256(b).hexdigest()})
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stream_digests.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
