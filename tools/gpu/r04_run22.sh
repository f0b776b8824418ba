cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'solve', d['phases_ms']['solve_total_last'], 'relerr %.1e' % d['relative_error'])"; }
for i in 1 2 3; do
HIPMF_EVENT_FENCE=1 run system_fence_events
HIPMF_EVENT_FENCE=0 run no_fence_events
done 2>&1 | tee gpurun_out/r04n/event_fence_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_matrix_zoo_gpu.py -m gpu -q -x 2>&1 | tail -4
