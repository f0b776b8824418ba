# rows of a long child fetched 1 / 2 (default) / 4 at a time per thread in the blocked forward slabs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05w
mkdir -p $OUT
for rep in 1 2; do
for lib in russell_amd/lib/variants/lib_u1.so "" russell_amd/lib/variants/lib_u4.so; do
echo "== lib=$lib rep $rep" >> $OUT/unroll.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/unroll.txt 2>&1
HIPMF_DEV_LIB=$lib timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/unroll.txt 2>&1
done
done
cat $OUT/unroll.txt
