#!/bin/bash
# Round 4, first GPU call: the GPU test-suite of the new tree, the forcing scan, and the bench at the
# per-GPU size of an 8-rank run (140 k tets) and at full size under the switches of this round's first experiments.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_properties_at_bench_size > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python tools/gpu_r4_forcing_scan.py > $O/forcing_scan.txt 2> $O/forcing_scan.err
rc=$?; echo "forcing scan rc=$rc"; cat $O/forcing_scan.txt
[ $rc -eq 124 ] && exit 1
run() {   # name tets steps warmup env...
  name=$1; tets=$2; steps=$3; warm=$4; shift 4
  env "$@" timeout -k 10 400 python bench.py --steps $steps --warmup $warm --tets $tets --no-cpu-baseline --no-fp64-line --profile-host > $O/$name.json 2> $O/$name.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]
    print("%-28s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app ortho %.1f spmv %.1f res %.1f host %s ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), pm["ortho_ms"], pm["spmv_ms"], pm["residual_ms"], {k: round(v,2) for k,v in (d.get("host_ms_per_step") or {}).items()}, {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s140_base        140000 12 3 A=1
run s140_jacobi      140000 12 3 FSI_VEL_JACOBI=1
run s140_schur4      140000 12 3 FSI_CHEB4=5
run s140_schur4_24   140000 12 3 FSI_CHEB4=5 FSI_CHEB_P=24
run s140_late        140000 12 3 FSI_NEWTON_FORCING_LATE=1e-3
run m1_base          1000000 20 5 A=1
run m1_jacobi        1000000 20 5 FSI_VEL_JACOBI=1
run m1_schur4        1000000 20 5 FSI_CHEB4=5
run m1_schur4_24     1000000 20 5 FSI_CHEB4=5 FSI_CHEB_P=24
run m1_late          1000000 20 5 FSI_NEWTON_FORCING_LATE=1e-3
run m1_late100       1000000 20 5 FSI_NEWTON_FORCING_LATE=1e-3 FSI_NEWTON_LATE_FACTOR=100
