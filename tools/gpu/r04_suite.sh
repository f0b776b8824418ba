# the driver's round-end sequence: all -m gpu tests, smoke, the default bench
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04s
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
( time timeout 900 python bench.py ) > $OUT/bench_default.txt 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.txt
