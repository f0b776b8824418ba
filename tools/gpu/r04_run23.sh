cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'solve', d['phases_ms']['solve_total_last'], 'relerr %.1e' % d['relative_error'])"; }
for i in 1 2 3; do run default; done 2>&1 | tee gpurun_out/r04n/la_mfma.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
