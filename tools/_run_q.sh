export VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "2 3" "4 2" "4 3" "4 4" "4 6"; do
set -- $cfg
VASPFSI_OVERLAP=$2 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $1 --steps 5 --warmup 0 --no-cpu-baseline --no-fp64-line > gpurun_out/r03_q_launch$1_ov$2.json 2> gpurun_out/r03_q_launch$1_ov$2.err
python tools/show_bench.py gpurun_out/r03_q_launch$1_ov$2.json | cut -c1-260
done
