cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06z4
mkdir -p $OUT
for rep in 1 2; do
for lib in "" russell_amd/lib/variants/lib_w1.so russell_amd/lib/variants/lib_c.so russell_amd/lib/variants/lib_e.so russell_amd/lib/variants/lib_f.so; do
echo "== lib=$lib rep $rep" >> $OUT/wt_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python tools/solve_variants.py 1000 only=defaults 2>&1 | grep -v "^matrix" | cut -c1-110 >> $OUT/wt_variants.txt
HIPMF_DEV_LIB=$lib timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   bench: value', d['value'], 'pair', d['phases_ms']['sptrsv_pair'], 'relative_error', d['relative_error'])" >> $OUT/wt_variants.txt
done
done
cat $OUT/wt_variants.txt
