#!/bin/bash
# Round 4, eighteenth GPU call: the solid cycle's coarse level measured (Lanczos numbers from CG steps at refresh time).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4r
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
run s_l100      140000 20 5 FSI_COARSE_LANCZOS=100 FSI_DEBUG=1
grep "Lanczos" $O/s_l100.err | head -2
run s_l200      140000 20 5 FSI_COARSE_LANCZOS=200 FSI_DEBUG=1
grep "Lanczos" $O/s_l200.err | head -2
run m_l100      1000000 20 5 FSI_COARSE_LANCZOS=100 FSI_DEBUG=1
grep "Lanczos" $O/m_l100.err | head -2
run m_l240      1000000 20 5 FSI_COARSE_LANCZOS=240 FSI_DEBUG=1
grep "Lanczos" $O/m_l240.err | head -2
