#!/bin/bash
mkdir -p gpurun_out/r2s31
t() { name=$1; shift; env FSI_DEBUG_PRECOND=1 "$@" timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "five_steps" -s > gpurun_out/r2s31/$name.log 2>&1; echo "$name rc=$?"; grep -E "precond\]|passed|failed|Error" gpurun_out/r2s31/$name.log | tail -12 | cut -c1-220; }
t fused
t fused_again
t unfused FSI_FUSED_SWEEPS=0
