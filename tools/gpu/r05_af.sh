cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05af
mkdir -p $OUT
for rep in 1 2 3; do
timeout 600 python tools/solve_variants.py 1000 only=defaults lib=russell_amd/lib/variants/lib_before.so >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=defaults >> $OUT/variants.txt 2>&1
done
grep -v "^matrix" $OUT/variants.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -2
