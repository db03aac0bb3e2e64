#!/bin/bash
# Round-4 evidence, ONE pass: parity suite (both lane mappings), smoke, bench (default; config5 GOP-batched and serial; --gpus 2 / 8 on the
# shared GPU, both workloads; config5 with the 1-rank RCCL communicator), rocprofv3 kernel stats of the default and the config5 workload,
# PMC passes (traffic, SQ groups, GRBM clock).  usage: bash tools/gpu_round4.sh <tag>
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
echo "== host: $(nproc) cpus, quota $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -m gpu -q --durations=12) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|^real" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $OUT/smoke.log
echo "== bench"; (time timeout 1200 python bench.py 2>$OUT/bench.err > $OUT/bench.json) 2>&1 | grep real; tail -2 $OUT/bench.err; python -c "
import json
r=json.load(open('$OUT/bench.json'))
print({k:r[k] for k in ('value','ms_per_step')}, 'bound', r['roofline']['bound'], 'frac', round(r['roofline']['frac'],4), 'traffic', r['roofline']['traffic'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()})
print('issue', r['roofline'].get('issue')); print('step', r['step_roofline']); print('sections', r['sections_s']); print('entropy', r.get('encode_to_payload',{}).get('value'))
print('extra', json.dumps(r.get('extra'))[:3000]); print('cpu', json.dumps(r.get('cpu_baseline'))[:600])"
echo "== bench --workload config5 (GOP-batched)"; timeout 900 python bench.py --workload config5 --steps 5 --warmup 2 2>$OUT/bench_c5.err > $OUT/bench_c5.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_c5.json')); print({k:r[k] for k in ('value','ms_per_step')}, 'frac', r['roofline']['frac'], 'traffic', r['roofline']['traffic'], {k:round(v['avg_launch_ms']*1e3,1) for k,v in r['kernels'].items()}); print(r['config']['workload'][:160]); print('step', r['step_roofline']); c=r['extra']['config4']; print(json.dumps(c['kernel_only'])[:500]); print({k:c['end_to_end'][k] for k in ('encode_value','decode_value','encoder_host_seconds','decoder_host_seconds')}, c['end_to_end']['serial_objects'])"
echo "== bench --workload config5 --serial-gops"; timeout 600 python bench.py --workload config5 --serial-gops --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-live-pmc 2>$OUT/bench_c5s.err > $OUT/bench_c5s.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_c5s.json')); print({k:r[k] for k in ('value','ms_per_step')}, 'frac', r['roofline']['frac'])"
echo "== bench --workload config5 --force-comm (1-rank RCCL communicator)"; timeout 600 python bench.py --workload config5 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-entropy --no-live-pmc --force-comm 2>$OUT/bench_c5_comm.err > $OUT/bench_c5_comm.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_c5_comm.json')); print({k:r[k] for k in ('value','rccl_ranks')}, r['control_plane']['backend'], r['control_plane']['ranks'])"
for N in 2 8; do
echo "== bench --gpus $N (one GPU shared: rendezvous + socket collectives)"; timeout 900 python bench.py --gpus $N --steps 3 --warmup 1 --streams $((96 / N)) --no-entropy 2>$OUT/bench_n$N.err > $OUT/bench_n$N.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_n$N.json')); print({k:r[k] for k in ('value','n_gpus','rccl_ranks')}, r['control_plane']['backend'], [(x['rank'], round(x['macroblocks_per_s']/1e6), x['device_ordinal'], x['pci_bus_id']) for x in r['control_plane']['ranks']])"
done
echo "== bench --gpus 8 --workload config5 (shared GPU)"; timeout 900 python bench.py --gpus 8 --workload config5 --frames 60 --steps 2 --warmup 1 --no-entropy 2>$OUT/bench_n8_c5.err > $OUT/bench_n8_c5.json; echo rc=$?; python -c "
import json
r=json.load(open('$OUT/bench_n8_c5.json')); print({k:r[k] for k in ('value','n_gpus','rccl_ranks')}, r['control_plane']['backend'], [(x['rank'], round(x['macroblocks_per_s']/1e6)) for x in r['control_plane']['ranks']])"
echo "== rocprofv3 kernel stats (default workload, then config5)"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-two-stream --no-extra --no-live-pmc > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -12 && cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o prof -- python $R/bench.py --workload config5 --steps 5 --warmup 2 --no-cpu-baseline --no-two-stream --no-extra --no-entropy --no-live-pmc > $OUT/prof_c5.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -8 && cp "$f" $OUT/kernel_stats_config5.csv
rm -rf $OUT/prof
echo "== PMC passes"
cd $R
bash $R/tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -f csv -d $OUT/clk -o clk -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-two-stream --no-extra --no-entropy --no-live-pmc > $OUT/clk.log 2>&1
f=$(find $OUT/clk -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $OUT/pmc/clk.summary.txt && grep "k_enc\|k_dec" $OUT/pmc/clk.summary.txt; k=$(find $OUT/clk -name "*kernel_trace.csv" | head -1); [ -n "$k" ] && python - "$k" >> $OUT/pmc/clk.summary.txt <<'PY'
import csv, re, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"pfv::(k_(?:enc|dec)_\w+)", r["Kernel_Name"])
    if m: d[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(f"{k:16s} duration_us              n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/clk
