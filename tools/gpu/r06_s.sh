cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06s
mkdir -p $OUT
timeout 1500 python tools/config4_knobs.py 200 "HIPMF_SPLIT_MINLEN=4096" "HIPMF_SPLIT_MINLEN=8192" "HIPMF_SPLIT_MINLEN=16384" "HIPMF_SPLIT_TASKS=0" "HIPMF_SPLIT_TASKS=256 HIPMF_SPLIT_MINLEN=4096" "HIPMF_SPLIT_TASKS=256 HIPMF_SPLIT_MINLEN=8192" "HIPMF_SPLIT_TASKS=128 HIPMF_SPLIT_MINLEN=8192" > $OUT/config4_knobs.txt 2>&1
cat $OUT/config4_knobs.txt
for ml in 2048 4096 8192; do
  echo "== HIPMF_SPLIT_MINLEN=$ml" >> $OUT/mid.txt
  HIPMF_SPLIT_MINLEN=$ml timeout 600 python tools/block_groups.py 3dl 144 64 4 >> $OUT/mid.txt 2>&1
  HIPMF_SPLIT_MINLEN=$ml timeout 600 python tools/block_groups.py 3d 100 64 4 >> $OUT/mid.txt 2>&1
done
cat $OUT/mid.txt | cut -c1-200
