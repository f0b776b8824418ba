cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ab
mkdir -p $OUT
for v in "HIPMF_MID_FWD_LEN5=64" "HIPMF_MID_FWD_LEN5=256" "HIPMF_MID_FWD_LEN5=64 HIPMF_MID_FWD_LEN6=16" "HIPMF_MID_FWD_LEN4=256" "HIPMF_MID_FWD_LEN6=64" "HIPMF_MID_FWD_LEN4=256 HIPMF_MID_FWD_LEN5=64"; do
timeout 600 python tools/solve_variants.py 1000 only=command $v >> $OUT/variants.txt 2>&1
done
timeout 600 python tools/solve_variants.py 1000 only=defaults >> $OUT/variants.txt 2>&1
grep -v "^matrix" $OUT/variants.txt
