cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
for v in a b c d; do
echo "== lib wt$v" >> gpurun_out/r03h/wt_lds_variants.txt
timeout 200 python tools/solve_variants.py 1000 lib=tools/ab/librussell_hipmf_wt$v.so "only=tree (defaults)" >> gpurun_out/r03h/wt_lds_variants.txt 2>&1
done
cat gpurun_out/r03h/wt_lds_variants.txt
