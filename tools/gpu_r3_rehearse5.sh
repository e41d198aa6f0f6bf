#!/bin/bash
# iteration counts of the partitioned solver at the bench size with FIVE ranks sharing the one card (gloo, host-staged):
# the closest rehearsal of the driver's 8-GPU run a one-GPU box allows (at most 6 processes on the card, and the launcher of torch.distributed.run counts as one).
# usage: tools/gpu_r3_rehearse5.sh   (writes gpurun_out/reh5/)
mkdir -p gpurun_out/reh5
export VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for ov in 3 4; do
  VASPFSI_OVERLAP=$ov timeout -k 10 500 python bench.py --gpus 5 --steps 5 --warmup 0 --no-cpu-baseline \
      > gpurun_out/reh5/launch5_ov$ov.json 2> gpurun_out/reh5/launch5_ov$ov.err
  rc=$?; echo "launch5 overlap $ov rc=$rc"
  python tools/show_bench.py gpurun_out/reh5/launch5_ov$ov.json | cut -c1-400
  [ $rc -eq 0 ] || exit $rc
done
