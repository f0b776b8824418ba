cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04h
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
# split trailing updates: same bits?  (threshold 1: every full step with a follower)
HIPMF_UPD_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
HIPMF_UPD_SPLIT=0 run split_off
HIPMF_UPD_SPLIT=3000 run split_3000
HIPMF_UPD_SPLIT=1000 run split_1000
HIPMF_UPD_SPLIT=300 run split_300
HIPMF_UPD_SPLIT=100 run split_100
done 2>&1 | tee gpurun_out/r04h/split_ab.txt
