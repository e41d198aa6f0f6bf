#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten.log 2>&1; echo "sten rc=$?"
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
for t in (50000, 250000): m = write_mesh('/tmp/mesh%d/stenosis.h5' % t, t); print(len(m['tets']))
"
timeout -k 10 200 python tools/gpu_lin.py offset_stenosis /tmp/mesh50000/stenosis.h5 0.001 0,1e-2,40 > gpurun_out/lin_50k.log 2>&1; echo "50k rc=$?"
timeout -k 10 300 python tools/gpu_lin.py offset_stenosis /tmp/mesh250000/stenosis.h5 0.001 0,1e-2,40 > gpurun_out/lin_250k.log 2>&1; echo "250k rc=$?"
tail -n 3 gpurun_out/lin_sten.log gpurun_out/lin_50k.log gpurun_out/lin_250k.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 25 gpurun_out/pytest_gpu.log
