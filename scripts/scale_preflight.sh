#!/bin/bash
# Pre-flight of an N-GPU node before `bench.py --gpus N` (scripts/scale_preflight.py says what it checks): scripts/scale_preflight.sh [N] [GB per rank]
cd "$(dirname "$0")/.."; N=${1:-8}; GB=${2:-88}
export HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${PORT:-29517} scripts/scale_preflight.py --gb $GB
