cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05x
mkdir -p $OUT
SOLVE_VARIANTS_SHORT=1 timeout 300 python tools/solve_variants.py 1000 > $OUT/variants.txt 2>&1
SOLVE_VARIANTS_SHORT=1 timeout 300 python tools/solve_variants.py 1000 >> $OUT/variants.txt 2>&1
cat $OUT/variants.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -3
