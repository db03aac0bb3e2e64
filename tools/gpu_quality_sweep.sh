#!/bin/bash
# Kernel times and rates of the default workload at several qualities (denser coefficients at higher quality: the entropy kernels'
# re-anchoring window, the scan's walk).  usage: bash tools/gpu_quality_sweep.sh <tag> [qualities...]
TAG=${1:-qsweep}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for q in ${@:-0 2 5 8 10}; do
  timeout 600 python bench.py --quality $q --steps 6 --warmup 2 --no-extra --no-cpu-baseline --no-two-stream 2>$OUT/q$q.err > $OUT/q$q.json || { echo "q$q failed"; tail -3 $OUT/q$q.err; continue; }
  python - $q $OUT/q$q.json <<'PY'
import json, sys
q, f = sys.argv[1:3]
r = json.load(open(f))
k = {n: round(v["avg_launch_ms"] * 1000, 1) for n, v in r["kernels"].items()}
e = r.get("encode_to_payload", {})
print(f"q={q:>2}: encode+decode {r['value'] / 1e9:.3f} G MB/s, coded {r['config'].get('pframe_coded_fraction')}, kernels us {k}, "
      f"encode_to_payload {e.get('value', 0) / 1e9:.3f} G ({e.get('ms_per_pass', 0):.2f} ms/pass), payload B/stream {e.get('last_pframe_payload_bytes_per_stream')}")
PY
done | tee $OUT/sweep.txt
