#!/bin/bash
# Round 4, twenty-first GPU call: the fluid predictor of the second chain starts when the solid cycle reaches its coarse level
# (experiment bit 2) instead of at once.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4u
mkdir -p $O
cd $R
run() {   # name tets steps warmup env...
  name=$1; tets=$2; steps=$3; warm=$4; shift 4
  env "$@" timeout -k 10 400 python bench.py --steps $steps --warmup $warm --tets $tets --no-cpu-baseline --no-fp64-line > $O/$name.json 2> $O/$name.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]
    print("%-26s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s140_x0   140000 12 3 FSI_EXPERIMENT=0
run s140_x4   140000 12 3 FSI_EXPERIMENT=4
run m1_x0     1000000 20 5 FSI_EXPERIMENT=0
run m1_x4     1000000 20 5 FSI_EXPERIMENT=4
