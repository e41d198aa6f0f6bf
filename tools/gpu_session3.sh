#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 python tools/gpu_lin.py cylinder tests/golden/cylinder/cylinder.h5 0.001 0,1e-2,40 > gpurun_out/lin_cyl.log 2>&1; echo "cyl rc=$?"
timeout -k 10 150 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten.log 2>&1; echo "sten rc=$?"
FSI_CHEB_S=1000 FSI_KAPPA_S=1e5 timeout -k 10 150 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten2.log 2>&1; echo "sten2 rc=$?"
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh50k/stenosis.h5', 50000); print(len(m['tets']))
"
timeout -k 10 200 python tools/gpu_lin.py offset_stenosis /tmp/mesh50k/stenosis.h5 0.001 0,1e-2,40 > gpurun_out/lin_50k.log 2>&1; echo "50k rc=$?"
tail -n 5 gpurun_out/lin_cyl.log gpurun_out/lin_sten.log gpurun_out/lin_sten2.log gpurun_out/lin_50k.log
