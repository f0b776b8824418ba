cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03final
for bs in 0 1; do
echo "== HIPMF_BLOCKED_SLABS=$bs"
HIPMF_BLOCKED_SLABS=$bs timeout 300 python tools/many_rhs.py 2d 1000 64 2>&1 | tail -1
HIPMF_BLOCKED_SLABS=$bs timeout 300 python tools/many_rhs.py 3d 100 64 2>&1 | tail -1
HIPMF_BLOCKED_SLABS=$bs timeout 600 python tools/config4_one_gpu.py 144 64 2>&1 | tail -1 | cut -c1-260
done
timeout 900 python tools/config4_one_gpu.py 200 256 > gpurun_out/r03final/config4_one_gpu.txt 2>&1
tail -1 gpurun_out/r03final/config4_one_gpu.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_round3_gpu.py tests/test_rccl_cabi_gpu.py -x -q 2>&1 | tail -3
