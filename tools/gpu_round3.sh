#!/bin/bash
# Round-3 evidence, ONE pass at the end of the round: parity suite (both lane mappings), smoke, bench (default; --gpus 2 on the shared
# GPU; config5 with and without the 1-rank RCCL communicator), rocprofv3 kernel stats, PMC passes.  usage: bash tools/gpu_round3.sh <tag>
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q --durations=12) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest_gpu.log | tail -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench"; timeout 1200 python bench.py 2>$OUT/bench.err > $OUT/bench.json; tail -2 $OUT/bench.err; python -c "
import json
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, 'frac', round(r['roofline']['frac'],4), {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()})
print('issue', r['roofline'].get('issue')); print('entropy', r.get('encode_to_payload',{}).get('value'))
print('extra', json.dumps(r.get('extra'))[:2500]); print('cpu', json.dumps(r.get('cpu_baseline'))[:600])"
echo "== bench --gpus 2 (one GPU shared: rendezvous + socket collectives)"; timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --streams 48 --no-entropy 2>$OUT/bench_n2.err > $OUT/bench_n2.json; echo rc=$?; cut -c1-300 $OUT/bench_n2.json
echo "== bench --workload config5"; timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_c5.json')); print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], json.dumps(r['extra']['config4']['kernel_only']))"
echo "== bench --workload config5 --force-comm"; timeout 600 python bench.py --workload config5 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-entropy --force-comm 2>$OUT/bench_c5_comm.err > $OUT/bench_c5_comm.json; echo rc=$?; cut -c1-200 $OUT/bench_c5_comm.json
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stream --no-extra > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -12 && cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/prof
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
