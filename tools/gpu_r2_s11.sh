#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s11
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > gpurun_out/r2s11/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r2s11/pytest.log | tail -8 | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2s11/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2s11/smoke.log | cut -c1-300
# N-rank launch path: bench.py starts 2 ranks itself (both on the one card of this box, host-staged transport)
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 VASPFSI_LIN_MAX_IT=600 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s11/launch2.json 2> gpurun_out/r2s11/launch2.err; echo "launch2 rc=$?"
python tools/show_bench.py gpurun_out/r2s11/launch2.json | cut -c1-500
bash tools/gpu_pmc_r2.sh 2>&1 | cut -c1-220
