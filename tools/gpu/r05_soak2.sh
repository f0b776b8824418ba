cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05soak2
timeout 1200 python tools/soak_big.py 20000 > gpurun_out/r05soak2/soak_big.txt 2>&1
cat gpurun_out/r05soak2/soak_big.txt
