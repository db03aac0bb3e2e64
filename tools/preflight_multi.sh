#!/bin/bash
# First-contact readiness of the N > 1 bench (round-5 review, item 4): both launch modes of `bench.py --gpus 8`.
#   HERE / on any box without 8 GPUs (CPU emulator, TCP control plane by design):   bash tools/preflight_multi.sh
#   on the 8-GPU node (real kernels, RCCL over xGMI; what the driver runs):         bash tools/preflight_multi.sh gpu
# The driver's own command is the third line of the "gpu" branch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
if [ "$1" = "gpu" ]; then
  N=${2:-8}
  export HSA_ENABLE_IPC_MODE_LEGACY=0
  echo "== self-launch"; python bench.py --gpus $N --steps 5 --warmup 2 --no-entropy | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_gpus','value','rccl_ranks','rccl_fallback','launch')}); assert d['rccl_ranks']==d['n_gpus'] and not d['rccl_fallback']"
  echo "== torchrun (the driver's command)"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 5 --warmup 2 --no-entropy | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print({k: d[k] for k in ('n_gpus','value','rccl_ranks','rccl_fallback','launch')}); assert d['rccl_ranks']==d['n_gpus'] and not d['rccl_fallback']"
else
  python tools/preflight_multi.py --ranks ${1:-8} --mode all
fi
