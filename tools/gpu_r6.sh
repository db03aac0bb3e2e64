#!/bin/bash
# Round 6, the evidence in ONE pass on one box.  usage: bash tools/gpu_r6.sh <tag> [soak minutes]
TAG=${1:-r06}; SOAK=${2:-10}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
tail -1 $OUT/build.log
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
echo "== bench, twice (cpu_baseline must agree between the two)"
for k in 1 2; do timeout 900 python bench.py 2>$OUT/bench$k.err > $OUT/bench$k.json; tail -2 $OUT/bench$k.err; done
python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
a, b = (json.load(open(f"{o}/bench{k}.json")) for k in (1, 2))
for d in (a, b):
    c = d["cpu_baseline"]
    print(f"value {d['value']/1e6:8.1f} M  k_enc_pframe {d['roofline']['avg_launch_ms']*1e3:6.1f} us frac {d['roofline']['frac']:.3f} issue {d['roofline']['issue']['frac_of_issue_floor']:.3f}"
          f"  cpu_baseline {c['value']/1e6:.3f} M on pool {c['threads_best']} of {c['cores']} cores, passes {c['passes_of_the_winner']}, p-frame encode {c['pframe_encode_value']/1e6:.3f} M  total {d['sections_s']['total']} s")
ca, cb = a["cpu_baseline"], b["cpu_baseline"]
print("cpu_baseline run-to-run: value %.1f %%, pframe_encode_value %.1f %%" % (100 * abs(ca["value"] - cb["value"]) / max(ca["value"], cb["value"]),
      100 * abs(ca["pframe_encode_value"] - cb["pframe_encode_value"]) / max(ca["pframe_encode_value"], cb["pframe_encode_value"])))
PY
echo "== bench --workload config5"; timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --no-extra 2>$OUT/bench_c5.err > $OUT/bench_c5.json; python -c "import json; d=json.load(open('$OUT/bench_c5.json')); print(d['value'], d['roofline']['frac'])"
echo "== rocprofv3 kernel stats, default workload"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stream --no-extra --no-live-pmc > $OUT/prof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
echo "== PMC passes"; bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log | cut -c1-300
echo "== soak, $SOAK min"; timeout $((SOAK * 60 + 300)) python tools/soak.py $SOAK 2>&1 | tail -4 | tee $OUT/soak.txt
