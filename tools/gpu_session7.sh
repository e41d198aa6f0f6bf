#!/bin/bash
mkdir -p gpurun_out
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh250k/stenosis.h5', 250000); print(len(m['tets']))
"
M=/tmp/mesh250k/stenosis.h5
timeout -k 10 300 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,40,100 150,3e3,20,100,40,100 100,1e3,20,100,40,100 60,1e3,10,30,20,100 150,3e3,10,30,20,30 > gpurun_out/tune_a.log 2>&1; echo "a rc=$?"
FSI_CHEB_D=40 FSI_KAPPA_D=400 timeout -k 10 300 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,40,100 150,3e3,20,100,40,100 > gpurun_out/tune_b.log 2>&1; echo "b rc=$?"
FSI_CHEB_D=15 FSI_KAPPA_D=50 timeout -k 10 300 python tools/gpu_tune.py offset_stenosis $M 0.001 150,3e3,20,100,40,100 > gpurun_out/tune_c.log 2>&1; echo "c rc=$?"
tail -n 7 gpurun_out/tune_a.log gpurun_out/tune_b.log gpurun_out/tune_c.log
