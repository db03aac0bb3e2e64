#!/bin/bash
# Round 6: the p-frame encoder as two kernels (k_pf_search + k_pf_transform, PFV_OPT_TILE_COMPACTION = 2) against the fused k_enc_pframe, on ONE
# box: (1) the p-frame parity tests under the split form, (2) per-kernel durations of every compiled variant of the two kernels
# (PFV_EXP_PSEARCH / PFV_EXP_PTRANSFORM select them at run time: one library, no rebuild) from rocprofv3 --kernel-trace --stats over a short
# run of the default bench, (3) the bench's own HIP-event figure (both kernels inside one pair of events) interleaved, ROUNDS times.
# usage: bash tools/gpu_split.sh <tag> [rounds] ["search variants"] ["transform variants"]
TAG=${1:-split}; ROUNDS=${2:-3}; SV=${3:-"0 1 2 3 4"}; TV=${4:-"0 1 2"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then
  echo "== parity, split form"
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lanes8split and not config5 and not config4" 2>&1 | tail -6 | tee $OUT/pytest_split.log
fi
BARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc"
prof() {  # name, env...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$name -o p -- python $R/bench.py $BARGS > $OUT/prof_$name.log 2>&1 )
  local f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/stats_$name.csv
  find $OUT/prof_$name -name "*kernel_trace.csv" -delete; find $OUT/prof_$name -name "*.db" -delete
}
echo "== per-kernel durations (rocprofv3 --kernel-trace --stats)"
prof fused PFV_BENCH_PENC_FORM=1
for s in $SV; do prof s${s} PFV_BENCH_PENC_FORM=2 PFV_EXP_PSEARCH=$s PFV_EXP_PTRANSFORM=1; done
for t in $TV; do prof t${t} PFV_BENCH_PENC_FORM=2 PFV_EXP_PSEARCH=0 PFV_EXP_PTRANSFORM=$t; done
python - $OUT <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/stats_*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if re.search(r"k_enc_pframe|k_pf_", r["Name"])]
    print(os.path.basename(f)[6:-4].ljust(8), "  ".join(f"{re.search(r'(k_enc_pframe|k_pf_search<[^>]*>|k_pf_transform<[^>]*>)', r['Name']).group(1)} {float(r['AverageNs'])/1e3:7.1f} us x{r['Calls']}" for r in rows))
PY
echo "== bench, HIP events around the encode call (both kernels), interleaved"
for r in $(seq 1 $ROUNDS); do
  PFV_BENCH_PENC_FORM=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc > $OUT/ab_fused.$r.json 2>$OUT/ab_fused.$r.err
  for s in $SV; do for t in $TV; do
    PFV_BENCH_PENC_FORM=2 PFV_EXP_PSEARCH=$s PFV_EXP_PTRANSFORM=$t timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc > $OUT/ab_s${s}t${t}.$r.json 2>$OUT/ab_s${s}t${t}.$r.err
  done; done
done
python - $OUT $ROUNDS <<'PY'
import glob, json, os, sys
out, rounds = sys.argv[1], int(sys.argv[2])
names = sorted({os.path.basename(f).split(".")[0] for f in glob.glob(out + "/ab_*.json")})
for n in names:
    cells = []
    for r in range(1, rounds + 1):
        try:
            d = json.load(open(f"{out}/{n}.{r}.json"))
            cells.append(f"{d['value']/1e6:7.1f} M  pe {d['kernels']['k_enc_pframe']['avg_launch_ms']*1e3:6.1f} us")
        except Exception:
            cells.append("failed".rjust(24))
    print(n[3:].ljust(10), "   ".join(cells))
PY
