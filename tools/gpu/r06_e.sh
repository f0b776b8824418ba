# round 6: block groups on LARGE factors (where do they stop paying?): 144^3 lower, config 4's matrix with one rank's shard
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/block_groups.py 3dl 144 64 1 2 4 > $OUT/block_groups_large.txt 2>&1
cat $OUT/block_groups_large.txt
for g in 1 2; do
  echo "== config 4 shard (200^3, 32 columns), HIPMF_BLOCK_GROUPS=$g" >> $OUT/block_groups_large.txt
  HIPMF_BLOCK_GROUPS=$g timeout 900 python tools/config4_one_gpu.py 200 32 >> $OUT/block_groups_large.txt 2>&1
done
tail -12 $OUT/block_groups_large.txt
