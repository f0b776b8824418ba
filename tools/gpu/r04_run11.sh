cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04i
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'solve', d['phases_ms'].get('solve'))"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2 3; do
HIPMF_BLOCK_INV=0 run binv_off
HIPMF_BLOCK_INV=1 run binv_on
done 2>&1 | tee gpurun_out/r04i/binv_ab.txt
