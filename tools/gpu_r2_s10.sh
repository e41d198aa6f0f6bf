#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s10
for v in "" "FSI_SPMV_COMPACT=0" "FSI_SCHUR_FP32=0"; do
  echo "=== $v"
  env $v timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 150 -k "properties_on_generated" > gpurun_out/r2s10/prop_"${v:-default}".log 2>&1; echo "rc=$?"
  grep -E "passed|failed|Error" gpurun_out/r2s10/prop_"${v:-default}".log | tail -2 | cut -c1-200
done
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2s10/$name.json 2> gpurun_out/r2s10/$name.err; echo "$name rc=$?"; python tools/show_bench.py gpurun_out/r2s10/$name.json | cut -c1-900; }
run compact_half
bash tools/gpu_pmc_r2.sh 2>&1 | cut -c1-220
