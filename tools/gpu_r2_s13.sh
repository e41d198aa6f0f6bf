#!/bin/bash
mkdir -p gpurun_out/r2s13
t() { name=$1; shift; sel=$1; shift; env FSI_DEBUG_PRECOND=1 "$@" timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 -k "$sel" -s > gpurun_out/r2s13/$name.log 2>&1; echo "$name rc=$?"; grep -E "precond\] self-test|passed|failed" gpurun_out/r2s13/$name.log | tail -4 | cut -c1-200; }
t isolated "properties"
t ka_then_prop "known_answer or properties"
t suite_schur64 "residual or jacobian or cylinder_three or known_answer or robin or properties" FSI_SCHUR_FP32=0
t robin_then_prop "robin or properties"
t cyl3_then_prop "cylinder_three or properties"
