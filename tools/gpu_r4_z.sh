#!/bin/bash
# Round 4, twenty-sixth GPU call: pressure rows of the outer product by k_spmv_prow - parity tests, bench at both sizes, kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_config3_size_with_the_robin_wall > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 124 ] && exit 1
run() {   # name tets steps warmup env...
  name=$1; tets=$2; steps=$3; warm=$4; shift 4
  env "$@" timeout -k 10 400 python bench.py --steps $steps --warmup $warm --tets $tets --no-cpu-baseline --no-fp64-line > $O/$name.json 2> $O/$name.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]; k=max(1,pc["precond_calls"])
    print("%-22s %7.2f it/s %6.1f ms/step newton %3d krylov %4d precond %.3f ortho %.3f spmv %.3f ms/product ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/k, pm["ortho_ms"]/k, pm["spmv_ms"]/max(1,pc["spmv_calls"]), {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run s140   140000 12 3 A=1
run m1     1000000 20 5 A=1
run m1_f64 1000000 20 5 FSI_KRYLOV_FP32=0 FSI_OPERATOR_FP32=0
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp64-line > $O/trace.json 2> $O/trace.err; echo "trace rc=$?"
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
grep -i "spmv" $O/kernel_stats.csv | cut -c1-200
