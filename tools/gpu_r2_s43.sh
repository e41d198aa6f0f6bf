#!/bin/bash
mkdir -p gpurun_out/r2s43
VASPFSI_FORCE_PARTITION=1 timeout -k 10 300 python -m cProfile -o gpurun_out/r2s43/prof.out bench.py --steps 2 --warmup 1 --tets 100000 --no-cpu-baseline > gpurun_out/r2s43/nccl1.json 2> gpurun_out/r2s43/nccl1.err; echo "rc=$?"
python - <<'PY'
import pstats
p = pstats.Stats('gpurun_out/r2s43/prof.out')
p.sort_stats('cumulative').print_stats(45)
PY
