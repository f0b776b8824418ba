# the driver's round-end sequence on the final commit of round 4
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f7
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f7/bench.json').read().strip().split('\n')[0])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d.get('speedup_repeat_call'), 'total_ifs', d.get('total_ifs_ms'), 'many', d['many_rhs']['solve_ms'], 'host', d['value_host_boundary_ms'])
PY
