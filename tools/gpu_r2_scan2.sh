#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/scan2
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/scan2/$name.json 2> gpurun_out/scan2/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/scan2/$name.json | head -1 | cut -c1-200; }
run f5 FSI_CHEB_F=5
run f6 FSI_CHEB_F=6
run f6k8 FSI_CHEB_F=6 FSI_KAPPA_F=8
run p30f5 FSI_CHEB_P=30 FSI_CHEB_F=5
run p30c FSI_CHEB_P=30 FSI_MG_CITS=30 FSI_SBMG_PRE=12 FSI_SBMG_POST=12
run p34 FSI_CHEB_P=34
