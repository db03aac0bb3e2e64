#!/bin/bash
# quick GPU pass: build, a few parity tests, the bench's kernel table.  usage: bash tools/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-quick}; KEXPR=${2:-"selfcheck or golden or trap or session_small or pframe_plane or hostile or benched"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu -k '$KEXPR'"; timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --no-extra --no-cpu-baseline --no-two-stream 2>$OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print("value %.4g  ms/step %.3f  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]))
for k, v in r["kernels"].items():
    print("  %-14s %8.1f us  %.3f of HBM peak" % (k, v["avg_launch_ms"] * 1e3, v["frac_of_hbm_peak"]))
e = r.get("encode_to_payload")
if e: print("  encode_to_payload %.4g MB/s (%.3f ms/pass)" % (e["value"], e["ms_per_pass"]))
PY
