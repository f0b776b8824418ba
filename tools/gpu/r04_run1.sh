# round 4, first measurement: k_front (one workgroup per mid-size front) against the tiled launches
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
for mid in 0 1; do
HIPMF_MID_FRONT=$mid timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_mid$mid.json 2> $OUT/bench_mid$mid.err
tail -c 1500 $OUT/bench_mid$mid.json
done
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
head -20 $OUT/kernel_stats.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fused_solve_gpu.py tests/test_matrix_zoo_gpu.py tests/test_random_patterns_gpu.py -m gpu -x -q > $OUT/pytest_a.txt 2>&1
tail -5 $OUT/pytest_a.txt
