python -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest_gpu_q.log 2>&1; tail -2 gpurun_out/r03_pytest_gpu_q.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line > $R/gpurun_out/r03_bench_q_prof.json 2> $R/gpurun_out/r03_bench_q_prof.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_q -name "*kernel_stats*.csv"); do cp $f gpurun_out/r03_q_kernel_stats.csv; done
python tools/show_bench.py gpurun_out/r03_bench_q_prof.json | cut -c1-330
