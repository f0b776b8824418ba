cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06y
mkdir -p $OUT
for setting in "X=0" "HIPMF_LEAF_KERNELS=0" "HIPMF_SOLVE_SLAB64=1" "HIPMF_SF_BIG_FRONT=512" "HIPMF_SF_BIG_FRONT=512 HIPMF_SF_BIG_ROWS=5" "HIPMF_SF_ASM_FRONT=1024" "HIPMF_SPLIT_TASKS=0" "HIPMF_BLOCK_COLS=8" "X=1"; do
  echo "== $setting" >> $OUT/c2_knobs.txt
  env $setting timeout 600 python tools/block_groups.py 2d 1000 256 4 2>&1 | cut -c1-170 >> $OUT/c2_knobs.txt
done
cat $OUT/c2_knobs.txt
