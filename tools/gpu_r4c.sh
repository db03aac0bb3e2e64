#!/bin/bash
# Round 4, third pass: the whole GPU suite, the default bench line, the 8-rank shared-GPU dry run of both workloads.
# usage: bash tools/gpu_r4c.sh <tag>
TAG=${1:-r04e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q --durations=12) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
echo "== bench (default)"; (time timeout 1200 python bench.py 2>$OUT/bench.err > $OUT/bench.json) 2>&1 | grep real; tail -2 $OUT/bench.err; python -c "
import json
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, 'bound', r['roofline']['bound'], 'frac', round(r['roofline']['frac'],4), {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()})
print('step', {k:r['step_roofline'][k] for k in ('frac','valu_issue_floor_ms','frac_of_valu_issue_floor')}); print('sections', r['sections_s'])
print('entropy', r.get('encode_to_payload',{}).get('value')); c=r['extra']['config4']; print('config4', json.dumps(c['kernel_only'])[:300], c['pcie_inclusive']['value'], {k:c['end_to_end'][k] for k in ('encode_value','decode_value')})
print('cpu', {k:r['cpu_baseline'][k] for k in ('value','cores','value_1thread')})"
echo "== bench --gpus 8 (one GPU shared: rendezvous + socket collectives), default workload, 12 streams per rank"; timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --streams 12 --no-entropy 2>$OUT/bench_n8.err > $OUT/bench_n8.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_n8.json')); print({k:r[k] for k in ('value','n_gpus','rccl_ranks')}, r['control_plane']['backend'], [ (x['rank'], round(x['macroblocks_per_s']/1e6), x['device_ordinal'], x['pci_bus_id']) for x in r['control_plane']['ranks']])"
echo "== bench --gpus 8 --workload config5 (shared GPU), 60-frame streams"; timeout 900 python bench.py --gpus 8 --workload config5 --frames 60 --steps 2 --warmup 1 --no-entropy 2>$OUT/bench_n8_c5.err > $OUT/bench_n8_c5.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_n8_c5.json')); print({k:r[k] for k in ('value','n_gpus','rccl_ranks')}, r['control_plane']['backend'], [ (x['rank'], round(x['macroblocks_per_s']/1e6)) for x in r['control_plane']['ranks']])"
