#!/bin/bash
# iteration counts of the partitioned solver at the bench size: 4 ranks sharing the one card (gloo, host-staged)
mkdir -p gpurun_out/reh
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 timeout -k 10 900 python bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/reh/launch4_1m.json 2> gpurun_out/reh/launch4_1m.err; echo "launch4 1M rc=$?"
python tools/show_bench.py gpurun_out/reh/launch4_1m.json | cut -c1-500
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/reh/single_5steps.json 2> gpurun_out/reh/single_5steps.err; echo "single rc=$?"
python tools/show_bench.py gpurun_out/reh/single_5steps.json | cut -c1-300
