cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ad
mkdir -p $OUT
for rep in 1 2; do
timeout 600 python tools/solve_variants.py 1000 only=command HIPMF_WT_SORT=0 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=defaults >> $OUT/variants.txt 2>&1
done
grep -v "^matrix" $OUT/variants.txt
