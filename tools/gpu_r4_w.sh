#!/bin/bash
# Round 4, twenty-third GPU call: the outlier modes of A M^-1 (tools/gpu_r4_outliers.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4w
cd $R
timeout -k 10 500 python tools/gpu_r4_outliers.py 48000 > gpurun_out/r4w/outliers_48k.txt 2> gpurun_out/r4w/outliers_48k.err
echo rc=$?; tail -44 gpurun_out/r4w/outliers_48k.txt | cut -c1-260; tail -3 gpurun_out/r4w/outliers_48k.err
