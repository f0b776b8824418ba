# round 5: pipelined leaf kernels
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_fused_solve_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_rccl_cabi_gpu.py -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
for leaf in 1 0 1 0; do
echo "HIPMF_LEAF_KERNELS=$leaf"
HIPMF_LEAF_KERNELS=$leaf timeout 300 python tools/many_rhs.py 2d 1000 64 2>&1 | tail -1
done > $OUT/many_rhs.txt 2>&1
cat $OUT/many_rhs.txt
cd /tmp && rm -rf /tmp/prof_many && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/many_rhs.py 2d 1000 64 0 > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) > $OUT/many_rhs_kernel_stats.txt 2>&1
grep -E "Li16|leaf" $OUT/many_rhs_kernel_stats.txt | cut -c1-130
