cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06z
mkdir -p $OUT
for rep in 1 2; do
for lib in "" russell_amd/lib/variants/lib_w1.so russell_amd/lib/variants/lib_w4.so russell_amd/lib/variants/lib_w1mi.so; do
echo "== lib=$lib rep $rep" >> $OUT/wt_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/solve_variants.py 1000 only=defaults 2>&1 | grep -v "^matrix" >> $OUT/wt_variants.txt
done
done
cat $OUT/wt_variants.txt
