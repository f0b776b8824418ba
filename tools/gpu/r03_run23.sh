cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03final
timeout 900 python tools/config4_one_gpu.py 200 256 > gpurun_out/r03final/config4_one_gpu.txt 2>&1
tail -1 gpurun_out/r03final/config4_one_gpu.txt
