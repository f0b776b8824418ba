cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05omega
timeout 300 python tools/omega_probe.py 2>&1 | grep -E "refinement step|column" > gpurun_out/r05omega/o.txt
cat gpurun_out/r05omega/o.txt
