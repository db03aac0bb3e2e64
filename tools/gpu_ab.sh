#!/bin/bash
# A/B of kernel variants on ONE box: each "NAME:-Dflags" is built on the box, then the default bench (kernel table only) runs on every
# variant in turn, ROUNDS times interleaved (boxes and minutes differ by a few per cent; the variants of one round are comparable).
# usage: bash tools/gpu_ab.sh <tag> <rounds> "NAME:-DFLAG ..." ...      (NAME "base" with empty flags = the tree as it is)
TAG=$1; ROUNDS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash tools/build_lib.sh /tmp/libpfv_$name.so -DPFV_BUILD_ID="\"ab-$name\"" $flags 2>$OUT/$name.build.err || { echo "$name: build failed"; tail -3 $OUT/$name.build.err; continue; }
  names+=($name)
done
for r in $(seq 1 $ROUNDS); do
  for name in "${names[@]}"; do
    PFV_HIP_LIB=/tmp/libpfv_$name.so timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-entropy --no-live-pmc $BENCH_ARGS > $OUT/$name.$r.json 2>$OUT/$name.$r.err || { echo "$name round $r failed"; tail -3 $OUT/$name.$r.err; }
  done
done
python - "$OUT" "$ROUNDS" "${names[@]}" <<'PY'
import sys, json
out, rounds, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
print(f"{'variant':16s} " + " ".join(f"{'round '+str(r):>34s}" for r in range(1, rounds + 1)))
for n in names:
    cells = []
    for r in range(1, rounds + 1):
        try:
            d = json.load(open(f"{out}/{n}.{r}.json"))
            k = d["kernels"]
            cells.append(f"{d['value']/1e6:7.1f} M  pe {k['k_enc_pframe']['avg_launch_ms']*1e3:6.1f} ie {k['k_enc_iframe']['avg_launch_ms']*1e3:5.1f} us")
        except Exception as e:
            cells.append(f"{'failed':>34s}")
    print(f"{n:16s} " + " ".join(f"{c:>34s}" for c in cells))
PY
