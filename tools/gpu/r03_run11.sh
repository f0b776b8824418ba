cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
for bc in 8 16; do
echo "== HIPMF_BLOCK_COLS=$bc" >> gpurun_out/r03k/many_rhs.txt
HIPMF_BLOCK_COLS=$bc timeout 300 python tools/many_rhs.py 2d 1000 64 >> gpurun_out/r03k/many_rhs.txt 2>&1
HIPMF_BLOCK_COLS=$bc timeout 300 python tools/many_rhs.py 3d 64 64 >> gpurun_out/r03k/many_rhs.txt 2>&1
done
cat gpurun_out/r03k/many_rhs.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_rccl_cabi_gpu.py -x -q 2>&1 | tail -3
