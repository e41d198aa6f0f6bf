#!/bin/bash
mkdir -p gpurun_out/r2s32
t() { name=$1; shift; env "$@" timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "five_steps" -s > gpurun_out/r2s32/$name.log 2>&1; echo "$name rc=$?"; grep -E "precond\]|passed|failed|Error" gpurun_out/r2s32/$name.log | tail -14 | cut -c1-260; }
t poison FSI_DEBUG_POISON=1 FSI_DEBUG_PRECOND=2
t poison_unfused FSI_DEBUG_POISON=1 FSI_DEBUG_PRECOND=2 FSI_FUSED_SWEEPS=0
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/r2s32/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2s32/pytest.log | cut -c1-200
