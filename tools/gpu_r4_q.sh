#!/bin/bash
# Round 4, seventeenth GPU call: sweeps and assumed condition number of the solid cycle's coarse level, both sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4q
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
    pm=d["phase_ms"]; pc=d["phase_calls"]; k=max(1,pc["precond_calls"])
    print("%-22s %7.2f it/s %6.1f ms/step newton %3d krylov %4d precond %.3f ortho %.3f spmv %.3f ms/it ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/k, pm["ortho_ms"]/k, pm["spmv_ms"]/k, {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s_40k1      140000 20 5 FSI_SBMG_CITS=40 FSI_SBMG_CKAPPA=1000
run s_30k500    140000 20 5 FSI_SBMG_CITS=30 FSI_SBMG_CKAPPA=500
run s_20k250    140000 20 5 FSI_SBMG_CITS=20 FSI_SBMG_CKAPPA=250
run s_30k1      140000 20 5 FSI_SBMG_CITS=30 FSI_SBMG_CKAPPA=1000
run s_40k2      140000 20 5 FSI_SBMG_CITS=40 FSI_SBMG_CKAPPA=2000
run s_40k1_mg32 140000 20 5 FSI_SBMG_CITS=40 FSI_SBMG_CKAPPA=1000 FSI_MG_CITS=32
run m_base      1000000 20 5 A=1
run m_60k2      1000000 20 5 FSI_SBMG_CITS=60 FSI_SBMG_CKAPPA=2000
run m_40k1      1000000 20 5 FSI_SBMG_CITS=40 FSI_SBMG_CKAPPA=1000
run m_30k500    1000000 20 5 FSI_SBMG_CITS=30 FSI_SBMG_CKAPPA=500
run m_20k250    1000000 20 5 FSI_SBMG_CITS=20 FSI_SBMG_CKAPPA=250
