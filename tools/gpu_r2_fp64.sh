#!/bin/bash
# the bench command with every storage-precision reduction of round 2 switched off (round 1's FP32 preconditioner sweeps stay)
mkdir -p gpurun_out/fp64
FSI_KRYLOV_FP32=0 FSI_OPERATOR_FP32=0 FSI_SCHUR_FP32=0 FSI_SWEEPS_FP16=0 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/fp64/bench.json 2> gpurun_out/fp64/bench.err; echo "rc=$?"
python tools/show_kernels.py gpurun_out/fp64/bench.json | cut -c1-200
