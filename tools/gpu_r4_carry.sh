#!/bin/bash
# Round 4: recycling across the Jacobian refresh (FsiTuning.krylov_carry): the bench with 0 / 32 / 64 / 128 carried directions.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4carry
mkdir -p $O
cd $R
for k in ${CARRY_LIST:-0 32 64 128}; do
  FSI_KRYLOV_CARRY=$k timeout -k 10 400 python bench.py --steps ${STEPS:-20} --warmup 5 --tets ${TETS:-1000000} --no-cpu-baseline --no-fp64-line > $O/carry_$k.json 2> $O/carry_$k.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/carry_$k.json") if l.startswith("{")][-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]; kk=max(1,pc["precond_calls"])
    print("carry %4d %7.2f it/s %6.1f ms/step newton %3d krylov %4d spmv calls %d factor+jac %.0f ms ev %s" % ($k, d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pc["spmv_calls"], pm["factor_ms"]+pm["jacobian_ms"], {a:b for a,b in d["solver_events"].items() if b}))
    print("     per step:", [sum(s) for s in d["krylov_per_solve"]])
except Exception as e:
    print("carry $k failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
done
