cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05match
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r05match/pytest_gpu.txt 2>&1
grep -E "passed|failed" gpurun_out/r05match/pytest_gpu.txt | tail -2
timeout 600 python tools/config3_standin.py > gpurun_out/r05match/config3.txt 2>&1; tail -6 gpurun_out/r05match/config3.txt | cut -c1-400
timeout 900 python tools/fuzz.py 200 13000 2>&1 | tail -1
timeout 900 python tools/fuzz_big.py 30 1500 2>&1 | tail -1
timeout 600 python tools/fuzz_complex_det.py 100 9500 2>&1 | tail -1
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) 2>&1 | grep -E "Max time spent on fact|Total time|fallbacks" | tr '\n' ' '
