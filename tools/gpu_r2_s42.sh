#!/bin/bash
mkdir -p gpurun_out/r2s42
VASPFSI_FORCE_PARTITION=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s42/nccl1.json 2> gpurun_out/r2s42/nccl1.err; echo "nccl one rank rc=$?"
tail -5 gpurun_out/r2s42/nccl1.err | cut -c1-300
python tools/show_bench.py gpurun_out/r2s42/nccl1.json | cut -c1-500
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s42/single.json 2> gpurun_out/r2s42/single.err; echo "single rc=$?"
python tools/show_bench.py gpurun_out/r2s42/single.json | cut -c1-300
