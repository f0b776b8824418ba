# thresholds of the split dot products / narrower forward slabs; stamps of a 16-column block at 144^3 with the default
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05r
mkdir -p $OUT
for t in 1024 2048 4096; do
echo "== HIPMF_SPLIT_TASKS=$t" >> $OUT/split.txt
HIPMF_SPLIT_TASKS=$t timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=$t timeout 900 python tools/config4_one_gpu.py 200 32 >> $OUT/split.txt 2>&1
done
echo "== HIPMF_SPLIT_TASKS=1024 HIPMF_SPLIT_MINLEN=1024" >> $OUT/split.txt
HIPMF_SPLIT_TASKS=1024 HIPMF_SPLIT_MINLEN=1024 timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=1024 HIPMF_SPLIT_MINLEN=1024 timeout 900 python tools/config4_one_gpu.py 200 32 >> $OUT/split.txt 2>&1
cut -c1-330 $OUT/split.txt
TRACE_3D=1 timeout 600 python tools/fused_trace_run.py $OUT/t.raw 144 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/t.raw > $OUT/solve_trace_144cube_16col.txt 2>&1
rm -f $OUT/t.raw
