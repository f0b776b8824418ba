cd $GRAFT_REPO_ROOT
for bc in 8 16; do
echo "== HIPMF_BLOCK_COLS=$bc"
HIPMF_BLOCK_COLS=$bc timeout 600 python tools/config4_one_gpu.py 144 64 2>&1 | tail -1
done
