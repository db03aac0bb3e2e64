#!/usr/bin/env python3
"""Markdown roofline table from one gpurun_out/<tag>/ directory (bench.json, prof/prof_kernel_stats.csv, pmc_traffic.json,
sq1.summary.txt): per kernel launch time, algorithmic vs measured HBM bytes, GB/s, fraction of the 8 TB/s roof, VALU
instructions and cycles per instruction per SIMD.  usage: python tools/roofline_table.py gpurun_out/<tag>"""
import csv
import json
import os
import re
import sys

def kname(full):
    """'void pfv::k_enc_pframe<true>(pfv::FrameGeom, ...)' -> 'k_enc_pframe' (template instances of one kernel are pooled)"""
    import re
    m = re.search(r"pfv::(k_\w+)", full)
    return m.group(1) if m else full


d = sys.argv[1]
bench = json.load(open(os.path.join(d, "bench.json")))
launch_mbs = bench["roofline"]["macroblocks_per_launch"]
algo = {"k_enc_iframe": 1024, "k_enc_pframe": 1284, "k_dec_iframe": 1024, "k_dec_pframe": 1284}   # B / macroblock, crop fused
stats = {}
for r in csv.DictReader(open(os.path.join(d, "prof", "prof_kernel_stats.csv"))):
    if "pfv::" in r["Name"]:
        stats[kname(r["Name"])] = float(r["AverageNs"]) / 1e3
# with the entropy stage on a second stream in half of the bench's extra passes, k_ent_* durations from rocprof include
# time-slicing; use their minimum-overlap figures from the same-stream pass when present
traffic = json.load(open(os.path.join(d, "pmc_traffic.json")))["kernels"]
sq = {}
for line in open(os.path.join(d, "sq1.summary.txt")):
    m = re.match(r"(\S+)\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
    if m:
        sq.setdefault(m.group(1), {})[m.group(2)] = float(m.group(3))
print("| kernel | µs / launch (rocprof avg) | algorithmic MB | HBM MB (PMC) | algorithmic GB/s | % of 8 TB/s | VALU wave-instr / launch | SIMD cycles per VALU instr |")
print("|---|---|---|---|---|---|---|---|")
for k in ("k_enc_pframe", "k_enc_iframe", "k_dec_pframe", "k_dec_iframe", "k_ent_scan", "k_ent_pack", "k_ent_codes", "k_ent_init"):
    if k not in stats:
        continue
    us = stats[k]
    a = launch_mbs * algo[k] / 1e6 if k in algo else None
    t = traffic.get(k, {}).get("traffic_bytes")
    valu = sq.get(k, {}).get("SQ_INSTS_VALU")
    gbs = (a or (t or 0) / 1e6) * 1e6 / (us * 1e-6) / 1e9
    # 256 CUs x 4 SIMDs, ~2.4 GHz
    cpi = us * 1e-6 * 2.4e9 * 1024 / valu if valu else None
    print(f"| `{k}` | {us:.1f} | {a:.0f} |" if a else f"| `{k}` | {us:.1f} | – |", end="")
    print(f" {t / 1e6:.0f} | {gbs:.0f}{'' if a else ' (measured bytes)'} | {gbs / 80:.1f} | {valu / 1e6:.1f} M | {cpi:.1f} |" if valu and t else " – | – | – | – | – |")
