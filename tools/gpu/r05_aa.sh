cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05aa
mkdir -p $OUT
timeout 600 python tools/solve_variants.py 1000 > $OUT/variants.txt 2>&1
grep -v "^matrix" $OUT/variants.txt
timeout 900 python -m pytest tests/test_fused_solve_gpu.py tests/test_round5_gpu.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/many_rhs.py 3d 100 16 2>&1 | tail -1
