cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04p
cd /tmp && rm -rf /tmp/prof3d && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof3d -o run -- python $GRAFT_REPO_ROOT/tools/run3d.py 100 lu > /tmp/prof3d.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof3d -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04p/kernel_stats_3d100_lu.txt 2>&1
head -14 gpurun_out/r04p/kernel_stats_3d100_lu.txt
tail -3 /tmp/prof3d.log
