cd $GRAFT_REPO_ROOT
for N in 160 176; do
HIPMF_BLOCK_COLS=16 timeout 900 python tools/config4_one_gpu.py $N 32 2>&1 | tail -1
done
