# config 4 in full on one GPU with independent random right-hand sides
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05n
mkdir -p $OUT
timeout 1200 python tools/config4_one_gpu.py 200 256 > $OUT/config4_one_gpu.txt 2>&1
tail -2 $OUT/config4_one_gpu.txt | cut -c1-800
