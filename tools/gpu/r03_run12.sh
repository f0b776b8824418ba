cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
timeout 1700 python -m pytest tests/test_round3_gpu.py -x -q --durations=8 2>&1 | tail -16 > gpurun_out/r03l/pytest_round3.txt
cat gpurun_out/r03l/pytest_round3.txt
