cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03i/pytest_gpu.txt
cat gpurun_out/r03i/pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err
tail -c 3000 gpurun_out/r03i/bench.json
