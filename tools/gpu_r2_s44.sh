#!/bin/bash
mkdir -p gpurun_out/r2s44
timeout -k 10 600 python -m pytest tests/test_gpu_partition.py -m gpu -q --timeout 400 -x > gpurun_out/r2s44/pytest_part.log 2>&1; echo "pytest partition rc=$?"
tail -5 gpurun_out/r2s44/pytest_part.log | cut -c1-250
VASPFSI_FORCE_PARTITION=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s44/nccl1.json 2> gpurun_out/r2s44/nccl1.err; echo "nccl one rank rc=$?"
python tools/show_bench.py gpurun_out/r2s44/nccl1.json | cut -c1-400
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 VASPFSI_LIN_MAX_IT=600 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s44/launch2.json 2> gpurun_out/r2s44/launch2.err; echo "launch2 rc=$?"
python tools/show_bench.py gpurun_out/r2s44/launch2.json | cut -c1-400
VASPFSI_FORCE_PARTITION=1 timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s44/nccl1_1m.json 2> gpurun_out/r2s44/nccl1_1m.err; echo "nccl one rank 1M rc=$?"
python tools/show_bench.py gpurun_out/r2s44/nccl1_1m.json | cut -c1-400
