#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_1m.json; tail -n 5 gpurun_out/bench_1m.err
