#!/bin/bash
# Final measurement of round 3, part B: PMC passes (calibration + the bench command), the 2-rank launch rehearsal on one card,
# the one-rank wire paths (torch nccl callbacks / the library's own RCCL calls), the 100 k-tet single context beside them.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
if [ -z "$SKIP_PMC" ]; then
bash tools/gpu_pmc_r3.sh 2>&1 | cut -c1-200
cp gpurun_out/pmc/pmc_traffic.json profiles/r03_pmc_traffic.json
fi
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 VASPFSI_LIN_MAX_IT=600 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/launch2.json 2> $O/launch2.err; echo "launch2 rc=$?"
python tools/show_bench.py $O/launch2.json | cut -c1-400
VASPFSI_FORCE_PARTITION=1 VASPFSI_RCCL=1 FSI_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/rccl_library_one_rank.json 2> $O/rccl_library_one_rank.err; echo "library rccl one rank rc=$?"
grep "collectives by the library" $O/rccl_library_one_rank.err | head -1
python tools/show_bench.py $O/rccl_library_one_rank.json | cut -c1-300
VASPFSI_FORCE_PARTITION=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/nccl_one_rank.json 2> $O/nccl_one_rank.err; echo "torch nccl one rank rc=$?"
python tools/show_bench.py $O/nccl_one_rank.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline --no-fp64-line > $O/single_100k.json 2> $O/single_100k.err; echo "single 100k rc=$?"
python tools/show_bench.py $O/single_100k.json | cut -c1-300
timeout -k 10 600 python bench.py --no-cpu-baseline --no-fp64-line --steps 100 --warmup 5 > $O/bench_100_steps.json 2> $O/bench_100_steps.err; echo "100 steps rc=$?"
python tools/show_bench.py $O/bench_100_steps.json | cut -c1-200
python tools/gpu_avf_case.py 50000 25 > $O/avf_50k_25steps.txt 2>&1; tail -1 $O/avf_50k_25steps.txt | cut -c1-250
python tools/gpu_avf_case.py 300000 12 > $O/avf_300k_12steps.txt 2>&1; tail -1 $O/avf_300k_12steps.txt | cut -c1-250
