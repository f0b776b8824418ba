cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05aj
timeout 300 python tools/host_solve_breakdown.py > gpurun_out/r05aj/b.txt 2>&1
cat gpurun_out/r05aj/b.txt
