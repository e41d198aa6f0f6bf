#!/bin/bash
# Round 4, fourth GPU call: two streams as the default - whole GPU test-suite, forcing scan on the new default, kernel trace
# with overlap statistics, and the bench with the candidate late forcing terms.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_bench_size > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
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
    print("%-28s %8.2f it/s %7.1f ms/step newton %3d krylov %4d precond %.3f ms/app ortho %.1f spmv %.1f res %.1f ev %s" % ("$name", d["value"], d["ms_per_step"], d["newton_iterations"], d["krylov_iterations"], pm["precond_ms"]/max(1,pc["precond_calls"]), pm["ortho_ms"], pm["spmv_ms"], pm["residual_ms"], {k:v for k,v in d["solver_events"].items() if v}))
except Exception as e:
    print("$name failed rc=$rc", e)
PY
  [ $rc -eq 124 ] && exit 1
}
run m1_default       1000000 20 5 A=1
run m1_late3         1000000 20 5 FSI_NEWTON_FORCING_LATE=3e-3
run m1_late2         1000000 20 5 FSI_NEWTON_FORCING_LATE=2e-3
run m1_late3_skip3   1000000 20 5 FSI_NEWTON_FORCING_LATE=3e-3 FSI_F32_VERDICT_SKIP=3e-4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_streams -- python3 $R/bench.py --steps 6 --warmup 2 --tets 1000000 --no-cpu-baseline --no-fp64-line > $O/prof_streams.json 2> $O/prof_streams.err
echo "rocprof rc=$?"
find /tmp/prof_streams -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/streams_kernel_stats.csv
find /tmp/prof_streams -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/trace_overlap.py {} > $O/streams_overlap.txt 2>&1
cat $O/streams_overlap.txt | tail -5
rm -rf /tmp/prof_streams
