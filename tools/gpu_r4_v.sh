#!/bin/bash
# Round 4, twenty-second GPU call: per-iteration log of the recycled GCR at 140 k tets (what happens around the Arnoldi steps
# and the forced second Gram-Schmidt passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4v
mkdir -p $O
cd $R
FSI_DEBUG_GCR=1 FSI_DEBUG_GCR_ALL=1 timeout -k 10 400 python bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line > $O/s140.json 2> $O/s140.err
echo rc=$?; grep -c "next from q" $O/s140.err; wc -l $O/s140.err
