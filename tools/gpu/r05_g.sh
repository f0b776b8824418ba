# round 5, seventh call: leaf kernels of the blocked solves
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_fused_solve_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_rccl_cabi_gpu.py tests/test_complex_twin_gpu.py tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
for leaf in 1 0 1 0; do
echo "HIPMF_LEAF_KERNELS=$leaf"
HIPMF_LEAF_KERNELS=$leaf HIPMF_BLOCK_COLS=16 timeout 300 python tools/many_rhs.py 2d 1000 64 2>&1 | tail -1
done > $OUT/many_rhs.txt 2>&1
HIPMF_LEAF_KERNELS=1 timeout 300 python tools/many_rhs.py 3d 100 64 2>&1 | tail -1 >> $OUT/many_rhs.txt
HIPMF_LEAF_KERNELS=0 timeout 300 python tools/many_rhs.py 3d 100 64 2>&1 | tail -1 >> $OUT/many_rhs.txt
cat $OUT/many_rhs.txt
for leaf in 1 0; do
HIPMF_LEAF_KERNELS=$leaf timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --grid3d 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('leaf kernels $leaf: many_rhs', d['many_rhs']['solve_ms'], d['many_rhs']['roofline']['ms_per_rhs'], d['many_rhs']['max_relative_error_all_columns'])"
done
