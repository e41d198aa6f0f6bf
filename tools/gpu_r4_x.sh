#!/bin/bash
# Round 4, twenty-fourth GPU call: forward elimination of the solid displacement residual in front of the block preconditioner,
# host prototype (tools/gpu_r4_forward_elim.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r4x
cd $R
timeout -k 10 800 python tools/gpu_r4_forward_elim.py 48000 > gpurun_out/r4x/forward_elim_48k.txt 2> gpurun_out/r4x/forward_elim_48k.err
echo rc=$?; cat gpurun_out/r4x/forward_elim_48k.txt | cut -c1-300; tail -3 gpurun_out/r4x/forward_elim_48k.err
