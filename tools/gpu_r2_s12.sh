#!/bin/bash
mkdir -p gpurun_out/r2s12
FSI_DEBUG_PRECOND=1 FSI_DEBUG_GCR=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "residual or jacobian or cylinder_three or known_answer or robin or properties" -s > gpurun_out/r2s12/suite.log 2>&1; echo "rc=$?"
grep -n "precond\] self-test\|PASSED\|FAILED\|passed\|failed\|^\[gcr\] it 10 \|no convergence" gpurun_out/r2s12/suite.log | tail -40 | cut -c1-230
