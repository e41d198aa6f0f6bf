#!/bin/bash
# Round 4, seventh GPU call: config-3-size property test, the 100-step run, the per-GPU size of an 8-rank run (140 k tets) with
# its kernel trace, and the PMC passes on the launch set of the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config3_size" > $O/pytest_config3.log 2>&1
rc=$?; echo "pytest config3 size rc=$rc"; grep -v "^$" $O/pytest_config3.log | tail -6
[ $rc -eq 124 ] && exit 1
timeout -k 10 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-fp64-line > $O/bench_100_steps.json 2> $O/bench_100_steps.err
rc=$?; echo "100 steps rc=$rc"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_100_steps.json") if l.startswith("{")][-1])
print("100 steps:", round(d["value"],2), "it/s", round(d["ms_per_step"],1), "ms/step newton", d["newton_iterations"], "krylov", d["krylov_iterations"], d["solver_events"])
for l in d.get("per_lifetime", []): print("   ", l)
PY
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line --profile-host > $O/bench_140k.json 2> $O/bench_140k.err
echo "140k rc=$?"; python tools/show_bench.py $O/bench_140k.json | cut -c1-420
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt140 -- python3 $R/bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line > $O/bench_140k_traced.json 2> $O/bench_140k_traced.err
echo "140k trace rc=$?"
find /tmp/kt140 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/small_140k_kernel_stats.csv
find /tmp/kt140 -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/trace_overlap.py {} > $O/small_140k_overlap.txt 2>&1
cat $O/small_140k_overlap.txt; rm -rf /tmp/kt140
cd $R
bash tools/gpu_pmc_r4.sh
