#!/bin/bash
# Round 4, third GPU call: the two-stream preconditioner application (FSI_PREC_STREAMS=1) - correctness on the parity tests
# that solve systems, then the bench at 140 k and 1.12 M tets against the single-stream form on the same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
FSI_PREC_STREAMS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "jacobian_spmv or cylinder_three or known_answer or five_steps or aneurysm_three or avf_two or properties_on_generated or production_storage" > $O/pytest_streams.log 2>&1
rc=$?; echo "pytest (FSI_PREC_STREAMS=1) rc=$rc"; tail -6 $O/pytest_streams.log
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
run s140_base      140000 12 3 A=1
run s140_streams   140000 12 3 FSI_PREC_STREAMS=1
run m1_base        1000000 20 5 A=1
run m1_streams     1000000 20 5 FSI_PREC_STREAMS=1
run m1_streams_b   1000000 20 5 FSI_PREC_STREAMS=1
cd /tmp && export TMPDIR=/tmp
FSI_PREC_STREAMS=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_streams -- python3 $R/bench.py --steps 6 --warmup 2 --tets 1000000 --no-cpu-baseline --no-fp64-line > $O/prof_streams.json 2> $O/prof_streams.err
echo "rocprof rc=$?"
find $O/prof_streams -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/streams_kernel_stats.csv
find $O/prof_streams -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/trace_overlap.py {} > $O/streams_overlap.txt 2>&1
cat $O/streams_overlap.txt | tail -20
rm -rf $O/prof_streams
