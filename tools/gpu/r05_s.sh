# split rule by length; run-to-run spread of config 4's shard
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05s
mkdir -p $OUT
for rep in 1 2 3; do
for t in 512 0; do
echo "== HIPMF_SPLIT_TASKS=$t rep $rep" >> $OUT/split.txt
HIPMF_SPLIT_TASKS=$t timeout 900 python tools/config4_one_gpu.py 200 32 2>&1 | grep -o '"solve_all_ms[^,]*' >> $OUT/split.txt
done
done
HIPMF_SPLIT_TASKS=512 timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=512 timeout 300 python tools/many_rhs.py 3d 144 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=0 timeout 300 python tools/many_rhs.py 3d 144 64 >> $OUT/split.txt 2>&1
cat $OUT/split.txt
