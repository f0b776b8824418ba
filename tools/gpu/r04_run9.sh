cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
for i in 1 2 3; do
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=192 HIPMF_MID_MMAX=1 run lu_192_only
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=192 run lu_192_plus_v3
HIPMF_MID_LU=0 run v3_only
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=128 HIPMF_MID_MMAX=1 run lu_128_only
done
