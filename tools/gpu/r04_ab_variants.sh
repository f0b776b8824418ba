# A/B of library variants under russell_amd/lib/variants against the default library, one call: VARIANTS="a b" bash r04_ab_variants.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
for rep in 1 2 3; do
for v in default $VARIANTS; do
if [ $v = default ]; then unset HIPMF_DEV_LIB; else export HIPMF_DEV_LIB=$GRAFT_REPO_ROOT/russell_amd/lib/variants/lib_$v.so; fi
timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v rep$rep value', d['value'], 'factor', d['phases_ms']['factor'], 'relerr %.1e' % d['relative_error'])"
done
done 2>&1 | tee gpurun_out/r04n/ab_variants.txt
