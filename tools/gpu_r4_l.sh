#!/bin/bash
# Round 4, twelfth GPU call: the A_dv product on the solid rows only and the 8-lane solid restriction - parity tests, then the
# bench at both sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_config3_size_with_the_robin_wall --deselect tests/test_gpu_parity.py::test_properties_at_bench_size > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 124 ] && exit 1
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
run s140     140000 12 3 A=1
run m1       1000000 20 5 A=1
run m1_b     1000000 20 5 A=1
