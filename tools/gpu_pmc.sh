#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group; --kernel-trace only).
# usage: bash tools/gpu_pmc.sh <tag> [bench args...]
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-stream --no-extra $BENCH_ARGS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" > $OUT/$name.summary.txt; cat $OUT/$name.summary.txt; cp "$f" $OUT/$name.counters.csv; rm -rf $OUT/$name; fi
}
BENCH_ARGS="$@"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run sq3 SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
cd $R; python $R/tools/pmc_traffic.py $OUT/fetch.counters.csv $OUT/write.counters.csv $OUT/pmc_traffic.json sq=$OUT/sq1.counters.csv streams=96 width=1920 height=1080 quality=5 build=$(python -c "import __graft_entry__ as g; print(g.hip_build_id())")
rm -f $OUT/*.counters.csv
