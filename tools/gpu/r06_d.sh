cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06d
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
timeout 600 python bench.py --no-configs --nrhs 0 --no-cpu-baseline > $OUT/bench_small.json 2> $OUT/bench_small.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d/bench_small.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'sym', d.get('symmetric', {}).get('value_ms'), 'host', d.get('value_host_boundary_ms'))
PY
