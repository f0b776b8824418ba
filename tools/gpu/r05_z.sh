cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05z
mkdir -p $OUT
timeout 600 python tools/solve_variants.py 1000 only=defaults HIPMF_MID_BWD_LEN4=256 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=command HIPMF_MID_BWD_LEN4=256 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=command HIPMF_MID_BWD_LEN4=128 HIPMF_MID_BWD_LEN5=64 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=command HIPMF_MID_BWD_LEN4=256 HIPMF_MID_BWD_LEN5=64 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=command HIPMF_MID_BWD_LEN4=1024 HIPMF_MID_BWD_LEN5=256 >> $OUT/variants.txt 2>&1
timeout 600 python tools/solve_variants.py 1000 only=defaults >> $OUT/variants.txt 2>&1
grep -v "^matrix" $OUT/variants.txt
