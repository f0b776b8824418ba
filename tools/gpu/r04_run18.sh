cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04l
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'assemble', d['phases_ms']['assemble'], 'solve', d['phases_ms']['solve_total_last'])"; }
for i in 1 2 3; do
HIPMF_OVERLAP_SMALL=0 run one_stream
run default
done 2>&1 | tee gpurun_out/r04l/pre_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_round2_gpu.py -m gpu -q -x 2>&1 | tail -3
