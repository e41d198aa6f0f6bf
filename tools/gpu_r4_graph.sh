#!/bin/bash
# Round 4: one application of the block preconditioner replayed from a captured HIP graph (FsiTuning.prec_graph) against the eager
# launches: parity tests with it, then the bench at three sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4graph
mkdir -p $O
cd $R
FSI_PREC_GRAPH=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or known_answer or bitwise or fixed_linear or production_storage" > $O/pytest.log 2>&1
rc=$?; echo "pytest (graph) rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 124 ] && exit 1
for t in 48000 140000 1000000; do for g in 0 1; do
  FSI_PREC_GRAPH=$g timeout -k 10 400 python bench.py --steps 12 --warmup 3 --tets $t --no-cpu-baseline --no-fp64-line > $O/t${t}_g$g.json 2> $O/t${t}_g$g.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/t${t}_g$g.json") if l.startswith("{")][-1])
    pm=d["phase_ms"]; pc=d["phase_calls"]; k=max(1,pc["precond_calls"])
    print("tets %8d graph %d %7.2f it/s %6.1f ms/step krylov %4d precond %.3f ortho %.3f spmv %.3f ms/it" % ($t, $g, d["value"], d["ms_per_step"], d["krylov_iterations"], pm["precond_ms"]/k, pm["ortho_ms"]/k, pm["spmv_ms"]/k))
except Exception as e:
    print("tets $t graph $g failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
done; done
