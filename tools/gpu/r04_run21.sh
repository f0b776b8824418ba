cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04m
cd /tmp && rm -rf /tmp/prof_ks && HIPMF_PAIR=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04m/kernel_stats_pair.txt 2>&1
python tools/factor_sequence.py $DB > gpurun_out/r04m/factor_sequence_pair.txt 2>&1
head -8 gpurun_out/r04m/kernel_stats_pair.txt
