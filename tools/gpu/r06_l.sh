# the very last build of round 6: GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06l
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06l/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'host', d.get('value_host_boundary_ms'))
print('many', d['many_rhs']['solve_ms'], d['many_rhs']['roofline']['ms_per_rhs'], 'config4', d['config4'].get('solve_s'), d['config4']['multi_gpu_model'].get('predicted_scaling_8_gpus'))
print('config5', d['config5'].get('ms_total'), d['config5'].get('fused_fallbacks'), d['config5'].get('gate_waits'))
print('speedups', d.get('speedup_one_shot'), d.get('speedup_repeat_call'))
PY
