#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s33
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s33/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s33/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s33/$name.json 2> gpurun_out/r2s33/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/r2s33/$name.json | cut -c1-200; }
run fp16
run fp32 FSI_SWEEPS_FP16=0
