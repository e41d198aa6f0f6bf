#!/bin/bash
# first GPU session of the round: parity dumps + the reference's known-answer run + a 50k-tet run
mkdir -p gpurun_out
timeout -k 10 240 python tools/gpu_debug.py cylinder > gpurun_out/debug_cyl.log 2>&1; echo "cyl rc=$?"
timeout -k 10 300 python tools/gpu_run_case.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0.04 > gpurun_out/pin_hip.log 2>&1; echo "pin rc=$?"
python -c "
import sys; sys.path.insert(0,'.')
from vasp_amd.meshgen import write_mesh
m = write_mesh('/tmp/mesh50k/stenosis.h5', 50000); print(len(m['tets']))
"
timeout -k 10 400 python tools/gpu_run_case.py offset_stenosis /tmp/mesh50k/stenosis.h5 0.001 0.003 > gpurun_out/run50k.log 2>&1; echo "50k rc=$?"
tail -n 30 gpurun_out/debug_cyl.log gpurun_out/pin_hip.log gpurun_out/run50k.log
