cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06ab
mkdir -p $OUT
for rep in 1 2; do
for setting in "X=0" "HIPMF_WT_FRONTS=16 HIPMF_WT_KB=48" "HIPMF_WT_FRONTS=12 HIPMF_WT_KB=32" "HIPMF_WT_FRONTS=32 HIPMF_WT_KB=96" "HIPMF_WT_FRONTS=24 HIPMF_WT_KB=48" "HIPMF_WT_FRONTS=8 HIPMF_WT_KB=24"; do
echo "== $setting rep $rep" >> $OUT/wt_caps.txt
timeout 300 python tools/solve_variants.py 1000 only=command $setting 2>&1 | grep -v "^matrix" | cut -c1-110 >> $OUT/wt_caps.txt
done
done
cat $OUT/wt_caps.txt
