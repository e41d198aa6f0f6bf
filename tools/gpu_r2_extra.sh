#!/bin/bash
# the two side measurements of profiles/README.md on the final state: all storage formats off; 2.6 M tets in one context
mkdir -p gpurun_out/extra
FSI_KRYLOV_FP32=0 FSI_OPERATOR_FP32=0 FSI_SCHUR_FP32=0 FSI_SWEEPS_FP16=0 FSI_PV_FP32=0 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/extra/fp64.json 2> gpurun_out/extra/fp64.err; echo "fp64 rc=$?"
python tools/show_kernels.py gpurun_out/extra/fp64.json | head -1
timeout -k 10 1000 python bench.py --tets 3000000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/extra/bench_3m.json 2> gpurun_out/extra/bench_3m.err; echo "3M rc=$?"
python tools/show_kernels.py gpurun_out/extra/bench_3m.json | head -1
