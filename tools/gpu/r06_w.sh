cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06w
mkdir -p $OUT
timeout 1500 python tools/config4_knobs.py 200 "HIPMF_BLOCKED_SLABS=1" "HIPMF_BLOCKED_SLABS=1 HIPMF_SF_BIG_ROWS=7" "HIPMF_BLOCKED_SLABS=1 HIPMF_SPLIT_MINLEN=16384" "HIPMF_BLOCKED_SLABS=1 HIPMF_SPLIT_TASKS=256" "HIPMF_BLOCKED_SLABS=1 HIPMF_BLOCK_GROUPS=1" > $OUT/config4_knobs.txt 2>&1
cat $OUT/config4_knobs.txt
for bs in 0 1; do
  echo "== HIPMF_BLOCKED_SLABS=$bs" >> $OUT/mid.txt
  HIPMF_BLOCKED_SLABS=$bs timeout 600 python tools/block_groups.py 2d 1000 256 4 >> $OUT/mid.txt 2>&1
  HIPMF_BLOCKED_SLABS=$bs timeout 600 python tools/block_groups.py 3d 100 64 4 >> $OUT/mid.txt 2>&1
  HIPMF_BLOCKED_SLABS=$bs timeout 600 python tools/block_groups.py 3dl 144 64 4 >> $OUT/mid.txt 2>&1
  HIPMF_BLOCKED_SLABS=$bs timeout 600 python tools/block_groups.py 2d 2000 64 4 >> $OUT/mid.txt 2>&1
done
cat $OUT/mid.txt | cut -c1-190
