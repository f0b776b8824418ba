cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05mk
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_many && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/many_rhs.py 2d 1000 64 0 > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) > gpurun_out/r05mk/many_rhs_kernel_stats.txt 2>&1
grep -E "Li16|leaf|cols" gpurun_out/r05mk/many_rhs_kernel_stats.txt | cut -c1-130
