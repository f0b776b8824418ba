cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 200 python tools/wt_stamps.py tools/ab/librussell_hipmf_stamps.so 1000 > gpurun_out/r03g/wt_stamps.txt 2>&1
tail -5 gpurun_out/r03g/wt_stamps.txt
