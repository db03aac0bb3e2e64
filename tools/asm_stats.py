#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels (hipcc -S): a quick VALU/LDS/VMEM census."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "pretty-fast-video_amd", "csrc", "pfv_capi.hip")
out = os.path.join(tempfile.gettempdir(), "pfv_asm.s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for f in re.split(r'\n(?=_ZN3pfv\w+:)', s):
    m = re.match(r'(_ZN3pfv\d+)(k_\w+?)E', f)
    if not m:
        continue
    body = f.split('s_endpgm')[0]
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(f"{m.group(2):16s} total {len(ins):5d} valu {grp('v_'):5d} salu {grp('s_'):4d} ds {grp('ds_'):4d} vmem {grp('global_') + grp('buffer_'):4d}")
    if len(sys.argv) > 1 and sys.argv[1] in m.group(2):
        print("   ", c.most_common(50))
