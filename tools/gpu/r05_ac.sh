cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ac
mkdir -p $OUT
for rep in 1 2 3; do
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) 2>&1 | grep -E "Max time spent on fact|Total time|fallbacks|real" | tr '\n' ' ' >> $OUT/c5.txt; echo >> $OUT/c5.txt
done
timeout 900 python tools/config4_one_gpu.py 200 16 2>&1 | grep -o '"solve_all_ms[^,]*' >> $OUT/c5.txt
for rep in 4 5; do
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) 2>&1 | grep -E "Max time spent on fact|Total time|fallbacks|real" | tr '\n' ' ' >> $OUT/c5.txt; echo >> $OUT/c5.txt
done
cat $OUT/c5.txt
