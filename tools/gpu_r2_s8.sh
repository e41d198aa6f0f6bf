#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s8
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > gpurun_out/r2s8/pytest.log 2>&1; echo "pytest fp64 rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r2s8/pytest.log | tail -8 | cut -c1-200
FSI_KRYLOV_FP32=1 timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > gpurun_out/r2s8/pytest_fp32.log 2>&1; echo "pytest fp32 rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r2s8/pytest_fp32.log | tail -8 | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s8/$name.json 2> gpurun_out/r2s8/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s8/$name.json | cut -c1-900; }
run schur32 FSI_SCHUR_FP32=1
run dbg FSI_DEBUG_GCR=1 
grep "^\[gcr\]" gpurun_out/r2s8/dbg.err | tail -4
run cap600 FSI_KRYLOV_CAP=600
