set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 300 python tools/solve_variants.py 1000 > gpurun_out/r03c/solve_variants_c2.txt 2>&1
cat gpurun_out/r03c/solve_variants_c2.txt
timeout 200 python tools/wt_stamps.py tools/ab/librussell_hipmf_stamps.so 1000 > gpurun_out/r03c/wt_stamps.txt 2>&1
tail -8 gpurun_out/r03c/wt_stamps.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r03c/pytest_fused.txt
cat gpurun_out/r03c/pytest_fused.txt
