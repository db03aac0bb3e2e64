#!/bin/bash
# Round-2 GPU sequence: parity suite, smoke, bench (default, --gpus 2 self-launch in shared-GPU mode, config5), kernel stats.
# usage (repo root on the GPU box): bash tools/gpu_round2.sh <tag>
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py "$@" 2>$OUT/bench.err > $OUT/bench.json; tail -3 $OUT/bench.err; python -c "
import json,sys
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()})
print('entropy', r.get('encode_to_payload',{}).get('value')); print('extra', json.dumps(r.get('extra'))[:1500]); print('cpu', json.dumps(r.get('cpu_baseline'))[:900])"
echo "== bench --gpus 2 (one GPU shared, gloo control plane)"; timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extra 2>$OUT/bench_n2.err > $OUT/bench_n2.json; echo rc=$?; tail -2 $OUT/bench_n2.err; cut -c1-400 $OUT/bench_n2.json
echo "== bench --workload config5"; timeout 600 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?; tail -2 $OUT/bench_c5.err; cut -c1-600 $OUT/bench_c5.json
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stream --no-extra "$@" > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" && cp "$f" $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
