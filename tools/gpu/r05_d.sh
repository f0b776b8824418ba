# round 5, fourth call: wave records with their first header, unarmed roots, mid stage 0
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_round5_gpu.py tests/test_round3_gpu.py -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 300 python tools/solve_variants.py 1000 > $OUT/solve_variants_c2.txt 2>&1
cat $OUT/solve_variants_c2.txt
