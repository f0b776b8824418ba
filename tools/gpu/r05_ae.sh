cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ae
mkdir -p $OUT
for rep in 1 2; do
timeout 600 python tools/solve_variants.py 1000 only=defaults lib=russell_amd/lib/variants/lib_norootgather.so >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=defaults >> $OUT/variants.txt 2>&1
done
grep -v "^matrix" $OUT/variants.txt
