#!/bin/bash
# capacity check: the bench workload at 3 M tets (BASELINE config 3 size) in ONE context
mkdir -p gpurun_out/big
timeout -k 10 1000 python bench.py --tets 3000000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/big/bench_3m.json 2> gpurun_out/big/bench_3m.err; echo "3M rc=$?"
python tools/show_kernels.py gpurun_out/big/bench_3m.json | cut -c1-200
tail -3 gpurun_out/big/bench_3m.err | cut -c1-200
