"""Static instruction counts of the kernels in a device-only assembly listing (hipcc -S --cuda-device-only).
usage: python tools/isa_count.py /tmp/capi.s [kernel-name-substring ...]   (writes /tmp/<kernel>.s for each match)"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r"^(_ZN3pfv\d+(k_\w+?)(?:ILb\d+E)?E\w*):.*?s_endpgm", s, re.S | re.M):
    name, body = m.group(2) + ("<%s>" % m.group(1).split("ILb")[1][0] if "ILb" in m.group(1) else ""), m.group(0)
    if want and not any(w in name for w in want):
        continue
    open("/tmp/%s.s" % name.replace("<", "_").replace(">", ""), "w").write(body)
    n = lambda pat: len(re.findall(r"^\s+" + pat, body, re.M))
    print(f"{name:20s} v_ {n('v_'):5d}  s_ {n('s_'):5d}  ds_ {n('ds_'):4d}  vmem {n('(global|buffer)_'):4d}")
