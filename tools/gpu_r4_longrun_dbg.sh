#!/bin/bash
# Round 4: the 100-step run with the per-cycle log of recurrence against true residual (FSI_DEBUG_TRUERES): what precedes a fall-back
# from the FP32 basis.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4long
mkdir -p $O
cd $R
FSI_DEBUG_TRUERES=1 timeout -k 10 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-fp64-line > $O/dbg.json 2> $O/dbg.err
echo rc=$?; python tools/show_bench.py $O/dbg.json | cut -c1-300; grep -c "fp32 cycle" $O/dbg.err; grep -n "basis fp64" $O/dbg.err | head -3
