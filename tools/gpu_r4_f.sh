#!/bin/bash
# Round 4, sixth GPU call: the refactored library (FsiTuning, four host translation units) - GPU test-suite incl. the full-size
# property test, the complete default bench line (with its FP64-storage child and the 48 k-tet CPU Krylov baseline: wall time
# of the whole command), and the aneurysm problem at its own tolerances under the new basis policy.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -eq 124 ] && exit 1
t0=$(date +%s)
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
rc=$?; t1=$(date +%s); echo "bench.py (default command) rc=$rc wall $((t1-t0)) s"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("value", round(d["value"],2), "ms/step", round(d["ms_per_step"],1), "fp64 storage", d.get("value_fp64_storage"), "newton", d["newton_iterations"], "krylov", d["krylov_iterations"])
print("roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k!="kernel"})
print("assembly_spmv", round(d["assembly_spmv"]["frac"],3), d["assembly_spmv"]["per_kernel_frac"])
c=d["cpu_baseline"]; print("cpu", round(c["value"],3), c["cores"], c["refresh_step"], c["steady_state"], c["phase_s"], "setup", c["setup_s"])
PY
[ $rc -eq 124 ] && exit 1
timeout -k 10 500 python tools/gpu_aneurysm_case.py 1000000 10 > $O/aneurysm_default.txt 2> $O/aneurysm_default.err
rc=$?; echo "aneurysm (own tolerances, default policy) rc=$rc"; tail -1 $O/aneurysm_default.txt
