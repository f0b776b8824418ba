cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05soak
mkdir -p $OUT
timeout 900 python tools/soak_big.py 3000 > $OUT/soak_big.txt 2>&1
cat $OUT/soak_big.txt
