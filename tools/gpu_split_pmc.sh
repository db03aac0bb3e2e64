#!/bin/bash
# SQ counter passes over the split p-frame encoder (k_pf_search / k_pf_transform, PFV_OPT_TILE_COMPACTION = 2) and the fused kernel beside it
# (one rocprofv3 run per counter group, --kernel-trace only).  usage: bash tools/gpu_split_pmc.sh <tag>
TAG=${1:-split_pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, form, counters...
  name=$1; form=$2; shift 2
  PFV_BENCH_PENC_FORM=$form timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-stream --no-extra --no-entropy --no-live-pmc > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py "$f" > $OUT/$name.summary.txt; grep -E "k_pf_|k_enc_pframe" $OUT/$name.summary.txt; rm -rf $OUT/$name; fi
}
for form in 2 1; do
run sq1_form$form $form SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2_form$form $form SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
done
run fetch_form2 2 FETCH_SIZE
run write_form2 2 WRITE_SIZE
