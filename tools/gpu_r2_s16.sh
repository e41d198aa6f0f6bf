#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s16
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > gpurun_out/r2s16/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r2s16/pytest.log | tail -8 | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s16/$name.json 2> gpurun_out/r2s16/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s16/$name.json | cut -c1-900; }
run op32
run op32_schur32 FSI_SCHUR_FP32=1
