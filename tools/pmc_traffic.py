#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE counter passes (rocprofv3 --pmc, separate runs) into per-kernel HBM bytes per launch.

usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json [sq=sq1_counter_collection.csv] key=value...
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
read stream (MI355X_MICROARCH.md, section HBM): it is doubled here; WRITE_SIZE is taken as is."""
import collections, csv, json, sys

def kname(full):
    """'void pfv::k_enc_pframe<true>(pfv::FrameGeom, ...)' -> 'k_enc_pframe' (template instances of one kernel are pooled)"""
    import re
    m = re.search(r"pfv::(k_\w+)", full)
    return m.group(1) if m else full


def means(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "pfv::" in k and row["Counter_Name"] == counter:
                acc[kname(k)].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
kv = dict(a.split("=", 1) for a in sys.argv[4:])
sq = kv.pop("sq", None)
valu = means(sq, "SQ_INSTS_VALU") if sq else {}
waves = means(sq, "SQ_WAVES") if sq else {}
out = {"note": "HBM bytes per launch = FETCH_SIZE[KiB]*1024*2 (gfx950 half-count correction) + WRITE_SIZE[KiB]*1024; "
               "valu_wave_instructions = SQ_INSTS_VALU per launch (issue cost per instruction: profiles/r02_ubench_valu_rates2.txt, r02_ubench_issue_patterns.txt)",
       "build_id": kv.pop("build", None),      # source hash of the libpfv_hip.so the counters were collected on (pfv_version())
       "config": kv, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    rd, wr = fetch.get(k, 0.0) * 1024 * 2, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr}
    if k in valu:
        out["kernels"][k].update({"valu_wave_instructions": valu[k], "wavefronts": waves.get(k)})
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
