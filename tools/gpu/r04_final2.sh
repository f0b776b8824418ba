# final state of round 4 (after the complex determinant): fuzz of the complex twin, the driver's round-end sequence, the headline profile
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/fuzz_complex_det.py 300 7000 > $OUT/fuzz_complex_det.txt 2>&1; tail -2 $OUT/fuzz_complex_det.txt
timeout 300 python tools/fuzz_host.py 60 > $OUT/fuzz_host.txt 2>&1; tail -1 $OUT/fuzz_host.txt
( time timeout 3000 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f2/bench.json').read().strip().split('\n')[0])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d.get('speedup_repeat_call'))
PY
