#!/bin/bash
# Two SQ counter passes (instruction counts, wait / LDS cycles) on the default bench, summaries only.  usage: bash tools/gpu_sq.sh <tag>
TAG=${1:-sq}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-stream --no-extra --no-entropy $BENCH_ARGS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" > $OUT/$name.summary.txt; grep "k_enc\|k_dec" $OUT/$name.summary.txt; rm -rf $OUT/$name; fi
}
BENCH_ARGS="$@"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
