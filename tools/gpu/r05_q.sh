# split dot products of the backward slabs + narrower forward slabs on levels of few tasks (blocked solves, large 3D factors): off / on
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05q
mkdir -p $OUT
for t in 0 512; do
echo "== HIPMF_SPLIT_TASKS=$t" >> $OUT/split.txt
HIPMF_SPLIT_TASKS=$t timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=$t timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/split.txt 2>&1
HIPMF_SPLIT_TASKS=$t timeout 900 python tools/config4_one_gpu.py 200 32 >> $OUT/split.txt 2>&1
done
cut -c1-420 $OUT/split.txt
( time timeout 900 python -m pytest tests -m gpu -q -x -k "blocked or many or blocks or leaf or config4 or tiny or rhs" ) > $OUT/pytest_subset.txt 2>&1
tail -5 $OUT/pytest_subset.txt
