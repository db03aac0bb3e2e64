#!/bin/bash
# Build kernel variants with -D switches on the GPU box and bench each (steps 4). usage: bash tools/gpu_variants.sh tag "NAME:-DFLAG ..." ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  lib=/tmp/libpfv_$name.so
  bash tools/build_lib.sh $lib $flags 2>$OUT/$name.build.err || { echo "$name: build failed"; tail -3 $OUT/$name.build.err; continue; }
  PFV_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/$name.json 2>$OUT/$name.err
  cd /tmp; export TMPDIR=/tmp
  PFV_HIP_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-entropy > /dev/null 2>&1
  cd $R
  python - "$name" "$OUT/$name.json" "$OUT/prof_$name/p_kernel_stats.csv" <<'PY'
import sys, json, csv
name, j, c = sys.argv[1:4]
try:
    d = json.load(open(j)); v = d["value"] / 1e6
except Exception as e:
    v = float("nan")
ks = {}
try:
    for r in csv.DictReader(open(c)):
        if "pfv::" in r["Name"]:
            import re
            ks[re.search(r"pfv::k_(\w+)", r["Name"]).group(1)] = round(float(r["AverageNs"]) / 1000, 1)
except Exception as e:
    ks = {"err": str(e)}
print(f"{name:14s} {v:8.1f} M MB/s  {ks}")
PY
  rm -rf $OUT/prof_$name
done
