#!/bin/bash
# Round 4, ninth GPU call: how much coupling the block factorisation needs between its chains (Krylov counts with the pressure
# right-hand side from the fluid predictor only / the displacement block without velocity coupling, one stream), and a 32-row
# Schur tile.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i
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
    print("%-24s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s140_default     140000 12 3 A=1
run s140_schur32     140000 12 3 FSI_SCHUR_TILE=32
run s140_exp1        140000 12 3 FSI_PREC_STREAMS=0 FSI_EXPERIMENT=1
run s140_exp2        140000 12 3 FSI_PREC_STREAMS=0 FSI_EXPERIMENT=2
run s140_exp3        140000 12 3 FSI_PREC_STREAMS=0 FSI_EXPERIMENT=3
run m1_default       1000000 20 5 A=1
run m1_schur32       1000000 20 5 FSI_SCHUR_TILE=32
run m1_exp1          1000000 20 5 FSI_PREC_STREAMS=0 FSI_EXPERIMENT=1
run m1_exp2          1000000 20 5 FSI_PREC_STREAMS=0 FSI_EXPERIMENT=2
