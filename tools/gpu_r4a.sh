#!/bin/bash
# Round 4, first pass: GOP-batched sessions on the GPU (parity vs the serial oracle) + config5 bench, GOP-batched vs serial.
# usage: bash tools/gpu_r4a.sh <tag>
TAG=${1:-r04a}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== pytest (new + touched paths)"; (time timeout 1500 python -m pytest tests -m gpu -q -x -k "native or gop_batched or bad_motion or session or batch_decoder or benched_shape or extreme_aspect" --durations=8) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest_gpu.log | tail -3
echo "== bench --workload config5 (GOP-batched)"; timeout 900 python bench.py --workload config5 --steps 3 --warmup 1 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?; tail -3 $OUT/bench_c5.err; python -c "
import json
r=json.load(open('$OUT/bench_c5.json')); print({k:r[k] for k in ('value','ms_per_step')}, 'frac', r['roofline']['frac'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()}); print(json.dumps(r['extra']['config4']['kernel_only'])); print('entropy', r.get('encode_to_payload',{}).get('value'))"
echo "== bench --workload config5 --serial-gops"; timeout 600 python bench.py --workload config5 --serial-gops --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-live-pmc 2>$OUT/bench_c5s.err > $OUT/bench_c5s.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_c5s.json')); print({k:r[k] for k in ('value','ms_per_step')}, 'frac', r['roofline']['frac'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()})"
echo "== GOPs per launch sweep (4K, kernel scope)"; timeout 600 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench, __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.Context(0)
for ngops in (1, 2, 4, 8, 12, 20):
    gb = bench.GopBatchSet(pkg, ctx, 3840, 2160, 5, [pkg.synth.SEED], ngops * 15)
    r = gb.wall(3); gb.verify(); gb.close()
    print(f"4K  {ngops:2d} GOPs per launch: {r/1e6:8.1f} M macroblocks/s", flush=True)
for ngops in (1, 4, 16, 40, 80):
    gb = bench.GopBatchSet(pkg, ctx, 1920, 1080, 5, [pkg.synth.SEED], ngops * 15)
    r = gb.wall(3); gb.verify(); gb.close()
    print(f"1080p {ngops:2d} GOPs per launch: {r/1e6:8.1f} M macroblocks/s", flush=True)
ctx.close()
PY
