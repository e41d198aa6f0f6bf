#!/bin/bash
mkdir -p gpurun_out/r2s7
for v in "" "FSI_SOLID_MG=0" "FSI_DD_MG=0" "FSI_SOLID_BJ=0" "FSI_SOLID_FP32=0"; do
  echo "=== $v" >> gpurun_out/r2s7/avf.log
  env FSI_DEBUG_PRECOND=1 $v timeout -k 10 120 python tools/gpu_debug_avf.py >> gpurun_out/r2s7/avf.log 2>&1
done
grep -E "===|precond|step" gpurun_out/r2s7/avf.log | cut -c1-250 | head -80
