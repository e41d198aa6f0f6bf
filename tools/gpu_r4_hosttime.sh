#!/bin/bash
# Round 4: how long the host takes to ISSUE one preconditioner application (FSI_DEBUG_PRECOND_HOST) against its duration on the device.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4host
mkdir -p $O
cd $R
for t in 48000 140000 1000000; do
  FSI_DEBUG_PRECOND_HOST=1 timeout -k 10 400 python bench.py --steps 12 --warmup 3 --tets $t --no-cpu-baseline --no-fp64-line > $O/t$t.json 2> $O/t$t.err
  echo "tets $t rc=$?"; grep "host time" $O/t$t.err | tail -1; python tools/show_bench.py $O/t$t.json | cut -c1-330
done
