#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 150 python tools/gpu_lin.py cylinder tests/golden/cylinder/cylinder.h5 0.001 0,1e-2,40 0,1e-3,300 0,1e-6,1000 > gpurun_out/lin_cyl.log 2>&1; echo "cyl rc=$?"
timeout -k 10 200 python tools/gpu_lin.py offset_stenosis tests/golden/offset_stenosis/offset_stenosis.h5 0.01 0,1e-3,300 0,1e-6,1000 > gpurun_out/lin_sten.log 2>&1; echo "sten rc=$?"
tail -n 12 gpurun_out/lin_cyl.log gpurun_out/lin_sten.log
