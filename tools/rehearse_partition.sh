#!/bin/bash
# N ranks of the partitioned bench on ONE card (host-staged gloo transport): iteration counts and correctness of the
# N > 1 path before the driver's multi-GPU run.  usage: tools/rehearse_partition.sh N TETS [STEPS]
set -e
N=${1:-2}; TETS=${2:-50000}; STEPS=${3:-3}
export VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus $N --steps $STEPS --warmup 1 --tets $TETS --no-cpu-baseline
