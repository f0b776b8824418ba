cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06z3
mkdir -p $OUT
for rep in 1 2; do
for lib in "" russell_amd/lib/variants/lib_w1mi.so russell_amd/lib/variants/lib_c.so; do
for prob in "100 3d" "2000" "500" "60 3d"; do
echo "== lib=$lib problem $prob rep $rep" >> $OUT/wt_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/solve_variants.py $prob only=defaults 2>&1 | grep -v "^matrix" >> $OUT/wt_variants.txt
done
done
done
cat $OUT/wt_variants.txt | cut -c1-125
# Radau5 config and the blocked solves with variant c
HIPMF_DEV_LIB=russell_amd/lib/variants/lib_c.so timeout 300 python tools/block_groups.py 2d 1000 256 4 2>&1 | cut -c1-170
