# round 5, first call: the tagged hand-offs on the device -- GPU tests, A/B of the solve schedules, trace, short bench
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
timeout 300 python tools/solve_variants.py 1000 > $OUT/solve_variants_c2.txt 2>&1
cat $OUT/solve_variants_c2.txt
timeout 120 python tools/fused_trace_run.py $OUT/trace_tag.txt 1000 > /dev/null 2>&1 && python tools/fused_trace.py $OUT/trace_tag.txt > $OUT/solve_trace_tag.txt 2>&1
cat $OUT/solve_trace_tag.txt
rm -f $OUT/trace_tag.txt
( time timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --nrhs 0 --grid3d 0 ) > $OUT/bench_short.json 2> $OUT/bench_short.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench_short.json').read().strip().split('\n')[0])
print('value', d['value'], d.get('phases_ms'), 'roofline', d['roofline'])
PY
