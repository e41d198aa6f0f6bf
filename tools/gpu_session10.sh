#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 python tools/gpu_lin.py cylinder tests/golden/cylinder/cylinder.h5 0.001 0,1e-2,40 > gpurun_out/lin_cyl.log 2>&1; echo "cyl rc=$?"
timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten.log 2>&1; echo "sten rc=$?"
FSI_SOLID_FP32=0 timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten64.log 2>&1; echo "sten64 rc=$?"
tail -n 2 gpurun_out/lin_cyl.log gpurun_out/lin_sten.log gpurun_out/lin_sten64.log
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh1m/stenosis.h5', 1000000); print(len(m['tets']))
"
M=/tmp/mesh1m/stenosis.h5
timeout -k 10 600 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,80,400,60,1000 300,1e4,20,100,40,100,60,1000 300,1e4,20,100,150,2000,60,1000 > gpurun_out/tune_1m.log 2>&1; echo "rc=$?"
tail -n 4 gpurun_out/tune_1m.log
