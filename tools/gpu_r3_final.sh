#!/bin/bash
# Final measurement of round 3 (one gpurun call): GPU tests, smoke, PMC passes (calibration + the bench command), the bench
# line (with value_fp64_storage and the one-lifetime CPU baseline), the same command under rocprofv3 --stats, the 2-rank
# launch rehearsal on one card and the one-rank wire path with the library's own RCCL calls.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -v --timeout 400 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "FAILED|ERROR|passed|failed" $O/pytest.log | tail -4 | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
bash tools/gpu_pmc_r3.sh 2>&1 | cut -c1-200
cp gpurun_out/pmc/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/show_kernels.py $O/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $R/bench.py --no-cpu-baseline --no-fp64-line > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "prof rc=$?"
cd $R
for f in $(find /tmp/prof_final -name "*kernel_stats*.csv"); do cp $f $O/kernel_stats.csv; done
python tools/show_kernels.py $O/bench_profiled.json | head -1 | cut -c1-200
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 VASPFSI_LIN_MAX_IT=600 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/launch2.json 2> $O/launch2.err; echo "launch2 rc=$?"
python tools/show_bench.py $O/launch2.json | cut -c1-400
VASPFSI_FORCE_PARTITION=1 VASPFSI_RCCL=1 FSI_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/rccl_library_one_rank.json 2> $O/rccl_library_one_rank.err; echo "library rccl one rank rc=$?"
grep "collectives by the library" $O/rccl_library_one_rank.err | head -1
python tools/show_bench.py $O/rccl_library_one_rank.json | cut -c1-300
VASPFSI_FORCE_PARTITION=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/nccl_one_rank.json 2> $O/nccl_one_rank.err; echo "torch nccl one rank rc=$?"
python tools/show_bench.py $O/nccl_one_rank.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline --no-fp64-line > $O/single_100k.json 2> $O/single_100k.err; echo "single 100k rc=$?"
python tools/show_bench.py $O/single_100k.json | cut -c1-300
