#!/bin/bash
# PMC passes over the decoder's device entropy stage (k_entd_*): VALU issue / clock, waits and LDS, HBM bytes.
#   usage: gpurun -- 'bash tools/gpu_entdec_pmc.sh'   (separate rocprofv3 runs per counter group, kernel trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/entdec_pmc
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
PFV_PROBE_COUNT_VALUES=1 PFV_PROBE_MODES="device->HBM" timeout 600 python $R/tools/entdec_probe.py 150 2>&1 | grep "packets type" | tee $OUT/values.txt
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  PFV_PROBE_MODES="device->HBM" timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python $R/tools/entdec_probe.py 150 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  k=$(find $OUT/$name -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" > $OUT/$name.summary.txt; grep "k_entd\|k_dec" $OUT/$name.summary.txt; [ -n "$k" ] && python - "$k" <<'PY' > $OUT/$name.durations.txt
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pfv::", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    print(f"{k:28s} n={len(v):4d} mean_us={sum(v) / len(v) / 1e3:9.1f} total_ms={sum(v) / 1e6:8.2f}")
PY
  grep "k_entd\|k_dec" $OUT/$name.durations.txt; rm -rf $OUT/$name; else tail -5 $OUT/$name.log; fi
}
run clk GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
# one file for profiles/: the counted values, the kernels' durations of each pass, the payload's value count
{ echo "== values (PFV_PROBE_COUNT_VALUES=1: what the stage delivers; the 150-frame probe decodes 10 packets per step: 1 step of i-frames, 14 of p-frames)"; cat $OUT/values.txt
  for n in clk wait fetch write; do echo "== $n (rocprofv3 --kernel-trace --pmc, tools/gpu_entdec_pmc.sh: tools/entdec_probe.py 150, frames left in HBM; 10 x 4K packets per step; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)"; grep "k_entd\|k_dec" $OUT/$n.summary.txt; grep "k_entd\|k_dec\|fillBuffer" $OUT/$n.durations.txt; done; } > $OUT/summary_all.txt
