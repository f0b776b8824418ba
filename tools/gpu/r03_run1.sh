set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
timeout 300 python tools/solve_variants.py 1000 > gpurun_out/r03a/solve_variants_c2.txt 2>&1
tail -12 gpurun_out/r03a/solve_variants_c2.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r03a/pytest_fused.txt
cat gpurun_out/r03a/pytest_fused.txt
timeout 300 python tools/solve_variants.py 100 3d > gpurun_out/r03a/solve_variants_3d100.txt 2>&1
tail -10 gpurun_out/r03a/solve_variants_3d100.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03a/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03a/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/r03a/bench_prof.log
DB=$(find gpurun_out/r03a/prof -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r03a/kernel_stats.txt 2>&1
head -30 gpurun_out/r03a/kernel_stats.txt
find gpurun_out/r03a/prof -name '*.db' -size +20M -delete
