#!/bin/bash
mkdir -p gpurun_out/r2s45
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "production_storage" -s > gpurun_out/r2s45/prec.log 2>&1; echo "precision test rc=$?"
grep -E "Newton iterations|passed|failed|Error|assert" gpurun_out/r2s45/prec.log | tail -8 | cut -c1-250
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2s45/launch2_1m.json 2> gpurun_out/r2s45/launch2_1m.err; echo "launch2 1M rc=$?"
python tools/show_bench.py gpurun_out/r2s45/launch2_1m.json | cut -c1-600
tail -3 gpurun_out/r2s45/launch2_1m.err | cut -c1-300
