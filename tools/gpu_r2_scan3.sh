#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/scan3
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/scan3/$name.json 2> gpurun_out/scan3/$name.err; echo "$name rc=$?"; python tools/show_kernels.py gpurun_out/scan3/$name.json | head -1 | cut -c1-200; }
run a FSI_MG_PRE=2 FSI_MG_POST=4 FSI_MG_ALPHA=10
run b FSI_MG_PRE=3 FSI_MG_POST=5 FSI_MG_ALPHA=16
run c FSI_MG_PRE=2 FSI_MG_POST=4 FSI_MG_ALPHA=10 FSI_MG_CITS=80 FSI_MG_CKAPPA=1000
run d FSI_MG_PRE=3 FSI_MG_POST=5 FSI_MG_ALPHA=16 FSI_MG_CITS=80 FSI_MG_CKAPPA=1000
run e FSI_MG_CITS=80 FSI_MG_CKAPPA=1000
run f FSI_MG_PRE=4 FSI_MG_POST=6 FSI_MG_ALPHA=20
run g FSI_SBMG_PRE=10 FSI_SBMG_POST=10 FSI_SBMG_ALPHA=80
run h FSI_SBMG_PRE=10 FSI_SBMG_POST=10 FSI_SBMG_ALPHA=80 FSI_SBMG_CITS=300 FSI_SBMG_CKAPPA=9000
