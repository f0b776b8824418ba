cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
for i in 1 2 3; do
HIPMF_EA_LDS=0 run ea_rmw
HIPMF_EA_LU=0 run ea_lds_only
HIPMF_EA_LU=1 run ea_lds_lu
done 2>&1 | tee gpurun_out/r04k/ea_lu_ab.txt
