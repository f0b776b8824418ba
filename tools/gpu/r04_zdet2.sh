# round 4, complex determinant: complex tests, leaf size of the paired ordering on config 5's K_comp, then the driver's round-end sequence
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04z
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_complex_twin_gpu.py tests/test_round2_gpu.py -m gpu -q ) > $OUT/pytest_complex.txt 2>&1
tail -5 $OUT/pytest_complex.txt
for leaf in 16 8 12; do
echo "HIPMF_ND_LEAF=$leaf (pairs)"
HIPMF_ND_LEAF=$leaf timeout 300 python tools/complex_breakdown.py 513 2>&1 | grep -v "^real" | tail -2
done > $OUT/complex_pairs_leaf.txt 2>&1
cat $OUT/complex_pairs_leaf.txt
( time timeout 3000 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
