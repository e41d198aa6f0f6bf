#!/bin/bash
mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.3f  ms/step %.0f  newton %d krylov %d  precond %.1f ms/apply  ortho %.0f spmv %.0f | %s %.3f ms' % (d['value'], d['ms_per_step'], d['newton_iterations'], d['krylov_iterations'], d['phase_ms']['precond_ms']/max(1,d['phase_calls']['precond_calls']), d['phase_ms']['ortho_ms'], d['phase_ms']['spmv_ms'], d['roofline']['kernel'][:24], d['roofline']['avg_launch_ms']))
"; }
run "FSI_X=0" | tee gpurun_out/sweep2.log
run "FSI_NO_TILES=1" | tee -a gpurun_out/sweep2.log
run "FSI_ORDER=mesh" | tee -a gpurun_out/sweep2.log
