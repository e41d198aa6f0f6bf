#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s25
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s25/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s25/pytest.log | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s25/$name.json 2> gpurun_out/r2s25/$name.err; echo "$name rc=$?"; grep "tiles:" gpurun_out/r2s25/$name.err | head -1; python tools/show_kernels.py gpurun_out/r2s25/$name.json | cut -c1-200; }
run split FSI_DEBUG_PRECOND=1
run split1024 FSI_TILE_THREADS=1024
run split256 FSI_TILE_THREADS=256
run nosplit FSI_NO_TILE_SPLIT=1
