#!/bin/bash
# round 2, session 1: parity tests with the new GCR, then bench variants (20 steps, warmup 5)
set -o pipefail
mkdir -p gpurun_out/r2s1
python -m pytest tests -m gpu -x -q > gpurun_out/r2s1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2s1/pytest.log
tail -5 gpurun_out/r2s1/pytest.log
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s1/$name.json 2> gpurun_out/r2s1/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s1/$name.json 2>/dev/null || tail -c 600 gpurun_out/r2s1/$name.json; }
run fp32_f1e-3 FSI_KRYLOV_FP32=1 FSI_NEWTON_FORCING=1e-3
run fp64_f1e-3 FSI_KRYLOV_FP32=0 FSI_NEWTON_FORCING=1e-3
run fp32_f1e-1 FSI_KRYLOV_FP32=1 FSI_NEWTON_FORCING=1e-1
run fp32_f3e-1 FSI_KRYLOV_FP32=1 FSI_NEWTON_FORCING=3e-1
