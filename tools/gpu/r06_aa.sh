cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06aa
mkdir -p $OUT
for rep in 1 2; do
for lib in "" russell_amd/lib/variants/lib_l1.so russell_amd/lib/variants/lib_l2.so russell_amd/lib/variants/lib_l1p4.so russell_amd/lib/variants/lib_l4p16.so; do
echo "== lib=$lib rep $rep" >> $OUT/leaf_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/block_groups.py 2d 1000 256 4 2>&1 | cut -c1-175 >> $OUT/leaf_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/block_groups.py 3d 100 64 4 2>&1 | cut -c1-175 >> $OUT/leaf_variants.txt
done
done
cat $OUT/leaf_variants.txt
