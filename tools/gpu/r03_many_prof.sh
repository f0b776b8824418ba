cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_many && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/ab/many_prof.py > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) 2>&1 | grep -E "fused|k_wt|perm|total kernel|calls"
