# SQ_INSTS_VALU of k_enc_pframe for ablated builds (instruction budget per section).  usage: bash tools/valu_ablate.sh "NAME:-DFLAGS" ...
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
R=$GRAFT_REPO_ROOT
names="full"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (cd pretty-fast-video_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $flags -o /tmp/lib_$name.so pfv_capi.hip) 2>/dev/null && names="$names $name"
done
cd /tmp && export TMPDIR=/tmp
for v in $names; do
  if [ $v = full ]; then unset PFV_HIP_LIB; else export PFV_HIP_LIB=/tmp/lib_$v.so; fi
  rm -rf /tmp/pmc_$v
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -f csv -d /tmp/pmc_$v -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-entropy --no-verify > /dev/null 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys, collections
name, f = sys.argv[1:3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_enc_pframe" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]); w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
print(f"{name:12s} VALU/launch {v/1e6:7.1f} M   per wavefront {v / w:7.1f}   per macroblock {v / (96 * 12240):6.1f}")
PY
done
