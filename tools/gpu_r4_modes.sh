#!/bin/bash
# Round 4: the GPU parity tests under the non-default storage / stream modes of FsiTuning (one stream; all-FP64 storage).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4modes
mkdir -p $O
cd $R
SEL="golden or known_answer or bitwise or fixed_linear or residual_matches or jacobian_spmv or robin or mooney or avf_two or aneurysm_three or device_diagnostics"
FSI_PREC_STREAMS=0 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$SEL" > $O/one_stream.log 2>&1; echo "one stream rc=$?"; tail -2 $O/one_stream.log
FSI_KRYLOV_FP32=0 FSI_OPERATOR_FP32=0 FSI_SCHUR_FP32=0 FSI_SWEEPS_FP16=0 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$SEL" > $O/fp64_storage.log 2>&1; echo "all-FP64 storage rc=$?"; tail -2 $O/fp64_storage.log
FSI_SWEEPS_FP32=0 FSI_SOLID_FP32=0 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden or known_answer or fixed_linear" > $O/fp64_sweeps.log 2>&1; echo "FP64 sweeps rc=$?"; tail -2 $O/fp64_sweeps.log
