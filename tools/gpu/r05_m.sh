# the leaf kernels on whole subtrees of small fronts (HIPMF_LEAF_TREE, default on) against the leaves only, blocked solves at C2 and 100^3
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
for t in 0 1; do
echo "== HIPMF_LEAF_TREE=$t" >> $OUT/many_rhs.txt
HIPMF_LEAF_TREE=$t timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/many_rhs.txt 2>&1
HIPMF_LEAF_TREE=$t timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/many_rhs.txt 2>&1
done
cat $OUT/many_rhs.txt
cd /tmp && rm -rf /tmp/prof_many && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/many_rhs.py 2d 1000 64 0 > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) > $OUT/many_rhs_kernel_stats.txt 2>&1
grep -E "Li16|leaf|cols" $OUT/many_rhs_kernel_stats.txt | cut -c1-140
( time timeout 900 python -m pytest tests -m gpu -q -x -k "blocked or many or blocks or leaf or config4 or tiny or rhs" ) > $OUT/pytest_subset.txt 2>&1
tail -5 $OUT/pytest_subset.txt
