for v in "X=1" "FSI_COARSE_POWER=0" "FSI_MG_PRE=4 FSI_MG_POST=6 FSI_MG_CITS=40" "FSI_MG_CITS=40" "VASPFSI_CELL_ORDER=mesh" "FSI_MG_PRE=4 FSI_MG_POST=6"; do
echo "== $v"
env $v python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "two_level_displacement" 2>&1 | grep -E "krylov iterations|passed|failed" | cut -c1-150
done
