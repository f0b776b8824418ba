# round 4: occupancy variants of the trailing-update kernels (library variants under russell_amd/lib/variants)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
export HIPMF_MID_FRONT=${MIDF:-0}
for v in default $VARIANTS; do
if [ $v = default ]; then unset HIPMF_DEV_LIB; else export HIPMF_DEV_LIB=$GRAFT_REPO_ROOT/russell_amd/lib/variants/lib_$v.so; fi
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v rep$rep value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'relerr %.1e' % d['relative_error'])"
done
done
