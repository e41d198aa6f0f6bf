#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s35
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s35/$name.json 2> gpurun_out/r2s35/$name.err; echo "$name rc=$?"; grep "tiles:" gpurun_out/r2s35/$name.err | head -1;  python tools/show_kernels.py gpurun_out/r2s35/$name.json | head -3 | cut -c1-200; }
run t128_256 FSI_DEBUG_PRECOND=1 FSI_TILE_THREADS=256
run t128_512 FSI_TILE_THREADS=512
