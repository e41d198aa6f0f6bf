#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s5
timeout -k 10 1000 python -m pytest tests -m gpu -v --timeout 400 -x > gpurun_out/r2s5/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2s5/pytest.log | tail -30 | cut -c1-200
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s5/$name.json 2> gpurun_out/r2s5/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s5/$name.json | cut -c1-900; }
run fp64
run fp64_f1e-2 FSI_NEWTON_FORCING=1e-2
run fp64_f3e-2 FSI_NEWTON_FORCING=3e-2
run fp32_f1e-2 FSI_KRYLOV_FP32=1 FSI_NEWTON_FORCING=1e-2
