#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -n 25 gpurun_out/pytest_gpu.log
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh50k/stenosis.h5', 50000); print(len(m['tets']))
"
timeout -k 10 300 python tools/gpu_run_case.py offset_stenosis /tmp/mesh50k/stenosis.h5 0.001 0.004 > gpurun_out/run50k.log 2>&1; echo "50k rc=$?"
tail -n 14 gpurun_out/run50k.log
