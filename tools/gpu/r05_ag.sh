cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05ag
python tools/microbench/host_copy_rates.py 40 > gpurun_out/r05ag/rates.txt 2>&1
python tools/microbench/host_copy_rates.py 8 >> gpurun_out/r05ag/rates.txt 2>&1
cat gpurun_out/r05ag/rates.txt
