#!/bin/bash
# one-at-a-time scan of the preconditioner's sweep counts around the defaults (bench workload)
set -o pipefail
mkdir -p gpurun_out/scan1
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/scan1/$name.json 2> gpurun_out/scan1/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/scan1/$name.json | head -1 | cut -c1-200; }
run base
run p30 FSI_CHEB_P=30
run p24 FSI_CHEB_P=24
run p30k60 FSI_CHEB_P=30 FSI_KAPPA_P=60
run mg36 FSI_MG_PRE=3 FSI_MG_POST=6
run mg24 FSI_MG_PRE=2 FSI_MG_POST=4
run mg44 FSI_MG_PRE=4 FSI_MG_POST=4
run mgc30 FSI_MG_CITS=30
run mgc20 FSI_MG_CITS=20
run sb12 FSI_SBMG_PRE=12 FSI_SBMG_POST=12
run sb8 FSI_SBMG_PRE=8 FSI_SBMG_POST=8
run sbc150 FSI_SBMG_CITS=150
run sbc100 FSI_SBMG_CITS=100
run f3 FSI_CHEB_F=3
run f2 FSI_CHEB_F=2
