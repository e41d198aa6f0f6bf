#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s17
env FSI_DEBUG_PRECOND=2 FSI_SCHUR_FP32=1 FSI_DEBUG_GCR=1 timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 150 -k properties -s > gpurun_out/r2s17/prop_schur32.log 2>&1; echo "prop rc=$?"
grep -E "precond\]|passed|failed" gpurun_out/r2s17/prop_schur32.log | head -30 | cut -c1-260
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s17/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s17/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s17/$name.json 2> gpurun_out/r2s17/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s17/$name.json | cut -c1-900; }
run op32
run op64 FSI_OPERATOR_FP32=0
