# round 4, complex determinant (paired pivot searches): the complex tests first, the cost of the paired mode on config 5's K_comp
# (HIPMF_COMPLEX_PAIRS=0: the plain real-equivalent factorisation), config 5 itself, then the driver's round-end sequence
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04z
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_complex_twin_gpu.py tests/test_round2_gpu.py -m gpu -x -q ) > $OUT/pytest_complex.txt 2>&1
tail -5 $OUT/pytest_complex.txt
for pairs in 1 0 1 0; do
echo "HIPMF_COMPLEX_PAIRS=$pairs"
HIPMF_COMPLEX_PAIRS=$pairs timeout 300 python tools/complex_breakdown.py 513 2>&1 | grep -v "^real"
done > $OUT/complex_pairs_cost.txt 2>&1
timeout 300 python tools/complex_breakdown.py 513 2>&1 | grep "^real" >> $OUT/complex_pairs_cost.txt
cat $OUT/complex_pairs_cost.txt
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) > $OUT/config5_radau5_brusselator_513.txt 2>&1
grep -i "time\|middle\|real" $OUT/config5_radau5_brusselator_513.txt
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
( time timeout 900 python bench.py ) > $OUT/bench_default.txt 2> $OUT/bench_default.err
tail -c 2500 $OUT/bench_default.txt
