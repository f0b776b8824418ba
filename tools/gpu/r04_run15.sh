# round 4: tile shapes of k_extend_add_lds (library variants under russell_amd/lib/variants)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
export TMPDIR=/tmp
for rep in 1 2; do
for v in default ea16x128 ea8x128 ea16x64 ea8x64 ea32x64; do
if [ $v = default ]; then unset HIPMF_DEV_LIB; else export HIPMF_DEV_LIB=$GRAFT_REPO_ROOT/russell_amd/lib/variants/lib_$v.so; fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$v rep$rep value', d['value'], 'factor', d['phases_ms']['factor'], 'relerr %.1e' % d['relative_error'])"
done
done 2>&1 | tee gpurun_out/r04k/ea_tiles.txt
