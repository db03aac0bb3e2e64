#!/usr/bin/env python3
"""One-off: the byte-flip fuzz of all three decoders (frame-by-frame, GOP-batched, batch) at 1280x720, where packets are big enough for
the device entropy stage under the DEFAULT option (the suite and tools/soak.py force it on small geometries).  usage: python tools/big_fuzz.py"""
import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tests')]
import __graft_entry__ as g
g.build_hip(); pkg=g.load_package()
import stream_cases as sc
from oracle_bind import Oracle
o=Oracle()
with pkg.Context(0) as ctx:
    w,h=1280,720
    data,_=sc.encode_pattern(pkg, ctx, o, w,h,5,"IPPPIPPDPIP", lambda buf: pkg.Encoder(buf,w,h,30,5,ctx), with_oracle=False)
    print(len(data), flush=True)
    print(sc.check_corrupted_streams(pkg, ctx, o, data, n_trials=24, seed=11), flush=True)
    print(sc.check_gop_decoder_corrupted(pkg, ctx, o, data, n_trials=24, seed=12), sc.ENTROPY_COUNTS, flush=True)
    sc.check_batch_decoder(pkg, ctx, o, 1280, 720, 5, n_streams=4, n_frames=4, gop=3)
    print("batch ok")
