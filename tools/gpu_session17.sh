#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.3f  ms/step %.0f  newton %d krylov %d  precond %.1f ms/apply  ortho %.0f spmv %.0f' % (d['value'], d['ms_per_step'], d['newton_iterations'], d['krylov_iterations'], d['phase_ms']['precond_ms']/max(1,d['phase_calls']['precond_calls']), d['phase_ms']['ortho_ms'], d['phase_ms']['spmv_ms']))
"; }
run "FSI_X=0" | tee gpurun_out/sweep.log
run "FSI_CHEB_D=30 FSI_KAPPA_D=300" | tee -a gpurun_out/sweep.log
run "FSI_CHEB_S=200 FSI_KAPPA_S=5000" | tee -a gpurun_out/sweep.log
run "FSI_CHEB_P=40 FSI_KAPPA_P=100" | tee -a gpurun_out/sweep.log
run "FSI_CHEB_D=30 FSI_KAPPA_D=300 FSI_CHEB_S=200 FSI_KAPPA_S=5000 FSI_CHEB_P=40 FSI_KAPPA_P=100" | tee -a gpurun_out/sweep.log
run "FSI_CHEB_D=20 FSI_KAPPA_D=100 FSI_CHEB_S=150 FSI_KAPPA_S=3000 FSI_CHEB_F=10 FSI_KAPPA_F=30" | tee -a gpurun_out/sweep.log
run "FSI_KRYLOV_CAP=150" | tee -a gpurun_out/sweep.log
