cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ai
HIPMF_EXP_NO_HOST_STAGE=1 timeout 600 python tools/host_boundary.py > gpurun_out/r05ai/host.txt 2>&1
timeout 600 python tools/host_boundary.py >> gpurun_out/r05ai/host.txt 2>&1
HIPMF_EXP_NO_HOST_STAGE=1 timeout 600 python tools/host_boundary.py >> gpurun_out/r05ai/host.txt 2>&1
grep -E "call [2-8]" gpurun_out/r05ai/host.txt
