# round 6: per-kernel stats of blocked solves with four groups per launch (C2, 64 + 256 columns), and the per-task trace of one super-block
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06b
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_many && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/block_groups.py 2d 1000 256 4 > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) > $OUT/many_rhs_kernel_stats_g4.txt 2>&1
cat /tmp/prof_many.log | tail -2
head -30 $OUT/many_rhs_kernel_stats_g4.txt
cd /tmp && rm -rf /tmp/prof_many1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many1 -o run -- python $GRAFT_REPO_ROOT/tools/block_groups.py 2d 1000 256 1 > /tmp/prof_many1.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many1 -name '*.db' | head -1) > $OUT/many_rhs_kernel_stats_g1.txt 2>&1
head -30 $OUT/many_rhs_kernel_stats_g1.txt
