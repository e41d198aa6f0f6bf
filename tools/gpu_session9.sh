#!/bin/bash
mkdir -p gpurun_out
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh1m/stenosis.h5', 1000000); print(len(m['tets']))
"
M=/tmp/mesh1m/stenosis.h5
timeout -k 10 900 python tools/gpu_tune.py offset_stenosis $M 0.001 300,1e4,20,100,40,100,15,50 300,1e4,20,100,40,100,60,1000 300,1e4,20,100,40,100,150,5000 300,1e4,10,30,20,30,150,5000 300,1e4,20,100,80,400,150,5000 150,3e3,20,100,40,100,150,5000 > gpurun_out/tune_1m.log 2>&1; echo "rc=$?"
tail -n 8 gpurun_out/tune_1m.log
