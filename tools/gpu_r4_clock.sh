#!/bin/bash
# Effective shader clock during each codec kernel (GRBM_GUI_ACTIVE / kernel duration) + the SQ busy / wait / issue counters, and the list of
# counters this gfx950 exposes.  usage: bash tools/gpu_r4_clock.sh <tag>
TAG=${1:-r04clk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1; grep -c "" $OUT/counters_avail.txt; grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|SQC_[A-Z_0-9]*" $OUT/counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-stream --no-extra --no-entropy --no-live-pmc > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  k=$(find $OUT/$name -name "*kernel_trace.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" > $OUT/$name.summary.txt; grep "k_enc\|k_dec" $OUT/$name.summary.txt; [ -n "$k" ] && cp "$k" $OUT/$name.kernel_trace.csv; cp "$f" $OUT/$name.counters.csv; rm -rf $OUT/$name; else tail -5 $OUT/$name.log; fi
}
run clk1 GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run clk2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
run clk3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES
