# round 5, second call: wave fronts + trimmed wave-subtree fetches; A/B against the forward instances compiled for five workgroups per CU
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_gpu_parity.py tests/test_round3_gpu.py -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
timeout 300 python tools/solve_variants.py 1000 > $OUT/solve_variants_c2.txt 2>&1
cat $OUT/solve_variants_c2.txt
SOLVE_VARIANTS_SHORT=1 timeout 200 python tools/solve_variants.py 1000 lib=$GRAFT_REPO_ROOT/russell_amd/lib/variants/lib_fwd5.so > $OUT/solve_variants_c2_fwd5.txt 2>&1
cat $OUT/solve_variants_c2_fwd5.txt
timeout 120 python tools/fused_trace_run.py $OUT/trace.txt 1000 > /dev/null 2>&1 && python tools/fused_trace.py $OUT/trace.txt > $OUT/solve_trace.txt 2>&1
cat $OUT/solve_trace.txt
rm -f $OUT/trace.txt
