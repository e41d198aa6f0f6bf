#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s30
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s30/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s30/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s30/$name.json 2> gpurun_out/r2s30/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/r2s30/$name.json | cut -c1-200; }
run dflt
