cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ah
timeout 600 python tools/host_boundary.py > gpurun_out/r05ah/host.txt 2>&1
timeout 600 python tools/host_boundary.py >> gpurun_out/r05ah/host.txt 2>&1
cat gpurun_out/r05ah/host.txt
timeout 600 python -m pytest tests/test_reference_api_gpu.py tests/test_host_mirror_gpu.py -m gpu -q 2>&1 | tail -2
