#!/bin/bash
# Round 4, thirteenth GPU call: additive two-level cycles (experiment bits 2 / 3: the coarse right-hand side from the INITIAL
# residual, so that the coarse chain could run beside the fine pre-sweeps) - sequential emulation, Krylov counts first.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m
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
run s140_base 140000 12 3 A=1
run s140_x4   140000 12 3 FSI_EXPERIMENT=4
run s140_x8   140000 12 3 FSI_EXPERIMENT=8
run s140_x12  140000 12 3 FSI_EXPERIMENT=12
run m1_base   1000000 20 5 A=1
run m1_x4     1000000 20 5 FSI_EXPERIMENT=4
run m1_x8     1000000 20 5 FSI_EXPERIMENT=8
run m1_x12    1000000 20 5 FSI_EXPERIMENT=12
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "config3 or bench_size or smoke or residual" > $O/pytest_config3.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_config3.log
