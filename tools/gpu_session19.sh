#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.3f  ms/step %.0f  newton %d krylov %d  precond %.1f ms/apply  ortho %.0f spmv %.0f' % (d['value'], d['ms_per_step'], d['newton_iterations'], d['krylov_iterations'], d['phase_ms']['precond_ms']/max(1,d['phase_calls']['precond_calls']), d['phase_ms']['ortho_ms'], d['phase_ms']['spmv_ms']))
"; }
run "FSI_SOLID_BJ=0" | tee gpurun_out/sweep3.log
run "FSI_SOLID_BJ=1" | tee -a gpurun_out/sweep3.log
run "FSI_CHEB_S=150 FSI_KAPPA_S=2000" | tee -a gpurun_out/sweep3.log
run "FSI_CHEB_S=120 FSI_KAPPA_S=1500" | tee -a gpurun_out/sweep3.log
run "FSI_CHEB_S=200 FSI_KAPPA_S=3000" | tee -a gpurun_out/sweep3.log
run "FSI_CHEB_S=100 FSI_KAPPA_S=1000" | tee -a gpurun_out/sweep3.log
timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten.log 2>&1; tail -n 1 gpurun_out/lin_sten.log
FSI_CHEB_S=150 FSI_KAPPA_S=2000 timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten2.log 2>&1; tail -n 1 gpurun_out/lin_sten2.log
