cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05t
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_c4 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o run -- python $GRAFT_REPO_ROOT/tools/config4_one_gpu.py 200 32 > $GRAFT_REPO_ROOT/$OUT/config4_shard.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_c4 -name '*.db' | head -1) > $OUT/config4_shard_kernel_stats.txt 2>&1
head -14 $OUT/config4_shard_kernel_stats.txt | cut -c1-150
grep -o '"solve_all_ms[^,]*' $OUT/config4_shard.txt | head -1
