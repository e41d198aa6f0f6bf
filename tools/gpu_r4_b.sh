#!/bin/bash
# Round 4, second GPU call: whole GPU test-suite (no -x), register-layout check of the MFMA Jacobian, the Jacobian variants in
# place at bench size (N1), and the early-displacement experiment.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mfma_layout_check.hip -o /tmp/mfma_layout_check && timeout -k 10 60 /tmp/mfma_layout_check > $O/layout_check.txt 2>&1
echo "layout check rc=$?"; cat $O/layout_check.txt
FSI_JAC_MFMA=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "jacobian_spmv or residual_matches or mooney or robin or avf_two or bitwise" > $O/pytest_mfma.log 2>&1
rc=$?; echo "pytest (FSI_JAC_MFMA=1) rc=$rc"; tail -4 $O/pytest_mfma.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python tools/gpu_r4_jacobian.py 1000000 > $O/jacobian_mfma.txt 2> $O/jacobian_mfma.err
rc=$?; echo "jacobian variants rc=$rc"; cat $O/jacobian_mfma.txt
[ $rc -eq 124 ] && exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_bench_size > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
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
    print("%-28s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app ortho %.1f spmv %.1f res %.1f ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), pm["ortho_ms"], pm["spmv_ms"], pm["residual_ms"], {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s140_ddearly       140000 12 3 FSI_DD_EARLY=1
run s140_ddearly_jac   140000 12 3 FSI_DD_EARLY=1 FSI_VEL_JACOBI=1
run m1_ddearly         1000000 20 5 FSI_DD_EARLY=1
run m1_ddearly_jac     1000000 20 5 FSI_DD_EARLY=1 FSI_VEL_JACOBI=1
run m1_late3           1000000 20 5 FSI_NEWTON_FORCING_LATE=3e-3
run m1_late_skip4      1000000 20 5 FSI_NEWTON_FORCING_LATE=1e-3 FSI_F32_VERDICT_SKIP=1e-4
