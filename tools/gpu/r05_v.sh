# packed gather of the children's vectors in the blocked forward slabs: variant library without it against the default build
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05v
mkdir -p $OUT
for rep in 1 2; do
for lib in russell_amd/lib/variants/lib_nopack.so ""; do
echo "== lib=$lib rep $rep" >> $OUT/pack.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/pack.txt 2>&1
HIPMF_DEV_LIB=$lib timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/pack.txt 2>&1
HIPMF_DEV_LIB=$lib timeout 900 python tools/config4_one_gpu.py 200 32 2>&1 | grep -o '"solve_all_ms[^,]*' >> $OUT/pack.txt
done
done
HIPMF_DEV_LIB=russell_amd/lib/variants/lib_nopack.so timeout 300 python tools/many_rhs.py 3d 144 64 >> $OUT/pack.txt 2>&1
timeout 300 python tools/many_rhs.py 3d 144 64 >> $OUT/pack.txt 2>&1
cat $OUT/pack.txt
