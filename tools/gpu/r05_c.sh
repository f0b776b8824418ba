# round 5, third call: the whole GPU suite (with the round-5 tests) and the full driver bench line (config3 / config4 / config5 sections)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -12 $OUT/pytest_gpu.txt | grep -v "^$"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('speedup_repeat_call'), 'total_ifs', d.get('total_ifs_ms'), 'host', d.get('value_host_boundary_ms'))
print('many', json.dumps(d.get('many_rhs'))[:600])
print('config3', json.dumps(d.get('config3'))[:900])
print('config4', json.dumps(d.get('config4'))[:900])
print('config5', json.dumps(d.get('config5'))[:700])
PY
