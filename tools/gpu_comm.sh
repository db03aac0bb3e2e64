#!/bin/bash
# RCCL control plane on the GPU box: the 1-rank communicator test, the 2-rank shared-GPU dry run of bench.py, N = 1 beside it.
TAG=${1:-comm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest rccl"; timeout 600 python -m pytest tests -m gpu -x -q -k "rccl" > $OUT/pytest_rccl.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_rccl.log | tail -3
echo "== bench --gpus 2 (two ranks sharing the one GPU: rendezvous + socket collectives)"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --streams 48 --no-entropy 2>$OUT/bench_gpus2.err > $OUT/bench_gpus2.json; echo rc=$?; tail -2 $OUT/bench_gpus2.err
python -c "
import json; r=json.load(open('$OUT/bench_gpus2.json')); print({k: r[k] for k in ('value','n_gpus','rccl_ranks','control_plane','ms_per_step')})"
echo "== bench --workload config5 N=1"
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --no-entropy --no-extra --no-cpu-baseline 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?
python -c "
import json; r=json.load(open('$OUT/bench_c5.json')); print({k: r[k] for k in ('value','n_gpus','ms_per_step')}, r['roofline']['frac'])"
echo "== bench --workload config5 N=1 --force-comm (1-rank RCCL communicator in the loop)"
timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 --no-entropy --no-extra --no-cpu-baseline --force-comm 2>$OUT/bench_c5_comm.err > $OUT/bench_c5_comm.json; echo rc=$?
python -c "
import json; r=json.load(open('$OUT/bench_c5_comm.json')); print({k: r[k] for k in ('value','n_gpus','ms_per_step','rccl_ranks')}, r['control_plane']['backend'])"
