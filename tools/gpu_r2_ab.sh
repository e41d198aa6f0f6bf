#!/bin/bash
# A/B of one switch on the bench, after the GPU tests
set -o pipefail
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/ab/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/ab/$name.json | cut -c1-200; }
run a
run b $AB_ENV
