cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
for i in 1 2 3; do
HIPMF_MID_LU_SPLIT=0 run one_launch
HIPMF_MID_LU_SPLIT=1 run by_lds_class
done 2>&1 | tee gpurun_out/r04l/mid_split_ab.txt
timeout 600 python -m pytest tests/test_round4_gpu.py -m gpu -q -x 2>&1 | tail -2
