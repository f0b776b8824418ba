cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03m/pytest_gpu.txt
cat gpurun_out/r03m/pytest_gpu.txt
