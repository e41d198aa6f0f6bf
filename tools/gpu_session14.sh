#!/bin/bash
mkdir -p gpurun_out/prof
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 6 gpurun_out/pytest_gpu.log
timeout -k 10 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; echo "bench rc=$?"
tail -c 2000 gpurun_out/bench_1m.json
