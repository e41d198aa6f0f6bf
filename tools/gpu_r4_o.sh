#!/bin/bash
# Round 4, fifteenth GPU call: how many of the Gram-Schmidt coefficients against the kept columns are significant
# (FSI_DEBUG_GCR prints the fractions above 1e-6 / 1e-9 / 1e-12 |w| every ten iterations).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4o
mkdir -p $O
cd $R
FSI_DEBUG_GCR=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --tets 1000000 --no-cpu-baseline --no-fp64-line > $O/m1_dbg.json 2> $O/m1_dbg.err
echo rc=$?; grep "stay below" $O/m1_dbg.err | tail -4
FSI_DEBUG_GCR=1 timeout -k 10 400 python bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line > $O/s140_dbg.json 2> $O/s140_dbg.err
echo rc=$?; grep "stay below" $O/s140_dbg.err | tail -4
