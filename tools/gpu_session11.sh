#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-2,40 > gpurun_out/lin_sten.log 2>&1; echo "sten rc=$?"
tail -n 2 gpurun_out/lin_sten.log
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh1m/stenosis.h5', 1000000); print(len(m['tets']))
"
M=/tmp/mesh1m/stenosis.h5
timeout -k 10 400 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,80,400,60,1000 > gpurun_out/tune_1m.log 2>&1; echo "rc=$?"
FSI_ORDER=colour timeout -k 10 400 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,80,400,60,1000 > gpurun_out/tune_1m_col.log 2>&1; echo "rc=$?"
tail -n 2 gpurun_out/tune_1m.log gpurun_out/tune_1m_col.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 6 gpurun_out/pytest_gpu.log
