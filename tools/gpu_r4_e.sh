#!/bin/bash
# Round 4, fifth GPU call: the GPU test-suite with the round's defaults (two streams, late forcing 3e-3, driver / worker ranks),
# a rehearsal of `bench.py --gpus 2` (driver / worker mode, both ranks on the one card over gloo) and the aneurysm problem at
# its own tolerances with the FP32 basis forced.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_properties_at_bench_size > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -eq 124 ] && exit 1
VASPFSI_DIST_BACKEND=gloo VASPFSI_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline > $O/launch2_driver.json 2> $O/launch2_driver.err
rc=$?; echo "bench --gpus 2 (driver / worker, gloo, one card) rc=$rc"; python tools/show_bench.py $O/launch2_driver.json | cut -c1-400; tail -3 $O/launch2_driver.err
[ $rc -eq 124 ] && exit 1
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --tets 100000 --no-cpu-baseline --no-fp64-line > $O/single_100k.json 2> $O/single_100k.err
echo "single 100k rc=$?"; python tools/show_bench.py $O/single_100k.json | cut -c1-300
for mode in default fp32; do
  if [ $mode = fp32 ]; then export FSI_KRYLOV_FP32=1; fi
  timeout -k 10 500 python tools/gpu_aneurysm_case.py 1000000 10 > $O/aneurysm_$mode.txt 2> $O/aneurysm_$mode.err
  rc=$?; echo "aneurysm $mode rc=$rc"; tail -2 $O/aneurysm_$mode.txt
  [ $rc -eq 124 ] && exit 1
done
unset FSI_KRYLOV_FP32
FSI_KRYLOV_FP32=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "five_steps_match or cylinder_three or aneurysm_three or aneurysm_production" > $O/pytest_fp32_forced.log 2>&1
echo "pytest with the FP32 basis forced rc=$?"; tail -5 $O/pytest_fp32_forced.log
