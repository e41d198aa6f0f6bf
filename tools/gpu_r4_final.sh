#!/bin/bash
# Round 4, final measurements of the committed tree: GPU test-suite, smoke, the default bench line, 140 k tets, then the PMC
# passes and the kernel trace (tools/gpu_pmc_r4.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
t0=$(date +%s)
timeout -k 10 900 python bench.py > $O/final_bench.json 2> $O/final_bench.err
rc=$?; t1=$(date +%s); echo "bench.py rc=$rc wall $((t1-t0)) s"
python tools/show_bench.py $O/final_bench.json | cut -c1-500
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --tets 140000 --no-cpu-baseline --no-fp64-line --profile-host > $O/small_140k_bench.json 2> $O/small_140k_bench.err
echo "140k rc=$?"; python tools/show_bench.py $O/small_140k_bench.json | cut -c1-400
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/launch2_driver.json 2> $O/launch2_driver.err
echo "bench --gpus 2 (driver / worker, gloo, one card) rc=$?"; python tools/show_bench.py $O/launch2_driver.json | cut -c1-300
bash tools/gpu_pmc_r4.sh
cd $R
timeout -k 10 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-fp64-line > $O/bench_100_steps.json 2> $O/bench_100_steps.err; echo "100 steps rc=$?"; python tools/show_bench.py $O/bench_100_steps.json | cut -c1-300
timeout -k 10 300 python tools/gpu_aneurysm_case.py 1000000 10 > $O/aneurysm_1m.txt 2>&1; echo "aneurysm rc=$?"; tail -1 $O/aneurysm_1m.txt | cut -c1-200
