#!/bin/bash
mkdir -p gpurun_out/r2s46
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 400 -k "bench_size" -s > gpurun_out/r2s46/big.log 2>&1; echo "bench-size property test rc=$?"
tail -6 gpurun_out/r2s46/big.log | cut -c1-250
