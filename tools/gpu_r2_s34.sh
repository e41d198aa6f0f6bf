#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2s34
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s34/$name.json 2> gpurun_out/r2s34/$name.err; echo "$name rc=$?"; grep "tiles:" gpurun_out/r2s34/$name.err | head -1;  python tools/show_kernels.py gpurun_out/r2s34/$name.json | head -3 | cut -c1-200; }
run t512_512 FSI_DEBUG_PRECOND=1
run t512_1024 FSI_TILE_THREADS=1024
