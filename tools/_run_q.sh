R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_q -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line --steps 22 --warmup 1 > $R/gpurun_out/r03_bench_q_prof.json 2> $R/gpurun_out/r03_bench_q_prof.err; echo "prof rc=$?"
cd $R
python tools/_trace_fill.py /tmp/prof_q
