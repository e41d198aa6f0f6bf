#!/bin/bash
# Round 4, twenty-fifth GPU call: driver / worker runs of the bench at 4 and 6 ranks on one card (host-staged gloo) and the
# same mesh in one context: Krylov counts and that the protocol holds with more than two ranks.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4y
mkdir -p $O
cd $R
export VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --tets 346000 --no-cpu-baseline --no-fp64-line > $O/one.json 2> $O/one.err; echo "1 context rc=$?"; python tools/show_bench.py $O/one.json | cut -c1-200
for n in 4 6; do
  timeout -k 10 500 python bench.py --gpus $n --steps 5 --warmup 1 --tets 346000 --no-cpu-baseline > $O/ranks$n.json 2> $O/ranks$n.err
  rc=$?; echo "$n ranks rc=$rc"; python tools/show_bench.py $O/ranks$n.json | cut -c1-330
  [ $rc -eq 124 ] && exit 1
done
