#!/bin/bash
# Round 4, tenth GPU call: with two chains side by side the pressure chain (stream B) has slack behind the solid / displacement
# chain (stream A) - do more Schur / fluid sweeps, which cost little wall time there, buy Krylov iterations?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4j
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
run m1_p30_f4       1000000 20 5 A=1
run m1_p40_f4       1000000 20 5 FSI_CHEB_P=40
run m1_p50_f4       1000000 20 5 FSI_CHEB_P=50
run m1_p50_k200     1000000 20 5 FSI_CHEB_P=50 FSI_KAPPA_P=200
run m1_p70_k300     1000000 20 5 FSI_CHEB_P=70 FSI_KAPPA_P=300
run m1_p30_f6       1000000 20 5 FSI_CHEB_F=6
run m1_p30_f8_k10   1000000 20 5 FSI_CHEB_F=8 FSI_KAPPA_F=10
run m1_p50_f6       1000000 20 5 FSI_CHEB_P=50 FSI_CHEB_F=6
run m1_p24_f4       1000000 20 5 FSI_CHEB_P=24
run s140_p30_f4     140000 12 3 A=1
run s140_p50_f4     140000 12 3 FSI_CHEB_P=50
run s140_p50_k200   140000 12 3 FSI_CHEB_P=50 FSI_KAPPA_P=200
