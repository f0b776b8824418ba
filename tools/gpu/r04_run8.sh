# round 4: k_front_lu (at most 32 pivots) in the factorisation, by the largest number of off-diagonal rows it takes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'launches', d['factor']['factor_launches'], 'relerr %.1e' % d['relative_error'])"; }
HIPMF_MID_LU=0 run lu_off
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=80 run lu_80
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=128 run lu_128
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=192 run lu_192
HIPMF_MID_LU=1 HIPMF_MID_LU_MMAX=192 HIPMF_MID_MMAX=1 run lu_192_only
HIPMF_MID_FRONT=0 run tiled_only
HIPMF_MID_LU=0 run lu_off_again
python -m pytest tests/test_round4_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
