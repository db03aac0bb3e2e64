#!/bin/bash
# Round 4, second pass: the GOP-batched objects (pfv_gop_encoder / pfv_gop_decoder) on the GPU + config #4 end to end through them.
# usage: bash tools/gpu_r4b.sh <tag> [pytest -k expression]
TAG=${1:-r04b}; K=${2:-gop}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== host: $(nproc) cpus, quota $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), mem:"; free -g | head -2
echo "== pytest -k '$K'"; (time timeout 2000 python -m pytest tests -m gpu -q -x -k "$K" --durations=8) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_gpu.log | head -20
echo "== bench --workload config5 (GOP-batched + config4 extras)"; timeout 1200 python bench.py --workload config5 --steps 3 --warmup 1 --no-live-pmc 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?; tail -5 $OUT/bench_c5.err; python -c "
import json
r=json.load(open('$OUT/bench_c5.json')); print({k:r[k] for k in ('value','ms_per_step')}, 'frac', r['roofline']['frac'])
c=r['extra']['config4']; print(json.dumps(c['kernel_only'])[:600]); print('pcie', c['pcie_inclusive']['value']); e=c['end_to_end']; print('e2e', {k:e[k] for k in ('encode_value','decode_value','value','upload_GBps_equivalent','parse_threads','stream_bytes')}, e['serial_objects'])"
