#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s19
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s19/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s19/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s19/$name.json 2> gpurun_out/r2s19/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s19/$name.json | cut -c1-900; }
run fused1024
run fused256 FSI_TILE_THREADS=256
run fused512 FSI_TILE_THREADS=512
run unfused FSI_FUSED_SWEEPS=0
