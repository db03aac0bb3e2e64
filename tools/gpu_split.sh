#!/bin/bash
# Round 6: the p-frame encoder as two kernels (k_pf_search + k_pf_transform, PFV_OPT_TILE_COMPACTION = 2) against the fused k_enc_pframe, on ONE
# box: (1) the p-frame parity tests under the split form, (2) per-kernel durations from rocprofv3 --kernel-trace --stats over a short run of the
# default bench, (3) the bench's own HIP-event figure (both kernels inside one pair of events), interleaved, ROUNDS times.
# usage: bash tools/gpu_split.sh <tag> [rounds]          (SKIP_TESTS=1: measurements only)
TAG=${1:-split}; ROUNDS=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then
  echo "== parity, split form"
  timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes8split" 2>&1 | tail -6 | tee $OUT/pytest_split.log
fi
BARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc"
prof() {  # name, form
  ( cd /tmp && export TMPDIR=/tmp && PFV_BENCH_PENC_FORM=$2 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$1 -o p -- python $R/bench.py $BARGS > $OUT/prof_$1.log 2>&1 )
  local f=$(find $OUT/prof_$1 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/stats_$1.csv
  rm -rf $OUT/prof_$1
}
echo "== per-kernel durations (rocprofv3 --kernel-trace --stats)"
prof fused 1
prof split 2
python - $OUT <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/stats_*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if re.search(r"k_enc_pframe|k_pf_", r["Name"])]
    print(os.path.basename(f)[6:-4].ljust(8), "  ".join(f"{re.search(r'(k_enc_pframe|k_pf_search|k_pf_transform)', r['Name']).group(1)} {float(r['AverageNs'])/1e3:7.1f} us x{r['Calls']}" for r in rows))
PY
echo "== bench, HIP events around the encode call (both kernels), interleaved"
for r in $(seq 1 $ROUNDS); do
  for form in 1 2; do
    PFV_BENCH_PENC_FORM=$form timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc > $OUT/ab_form$form.$r.json 2>$OUT/ab_form$form.$r.err
  done
done
python - $OUT $ROUNDS <<'PY'
import glob, json, os, sys
out, rounds = sys.argv[1], int(sys.argv[2])
for n, label in (("ab_form1", "fused (tile compaction)"), ("ab_form2", "split (search + transform)")):
    cells = []
    for r in range(1, rounds + 1):
        try:
            d = json.load(open(f"{out}/{n}.{r}.json"))
            cells.append(f"{d['value']/1e6:7.1f} M  pe {d['kernels']['k_enc_pframe']['avg_launch_ms']*1e3:6.1f} us")
        except Exception:
            cells.append("failed".rjust(24))
    print(label.ljust(28), "   ".join(cells))
PY
