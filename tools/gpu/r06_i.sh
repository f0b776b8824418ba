cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06i
mkdir -p $OUT
( echo "two processes on one GPU, 1000 x 1000 Poisson, 1500 solves each (tools/soak_two_processes.py):"
  timeout 300 python tools/soak_two_processes.py 1000 1500
  echo "the same with HIPMF_PROCESS_GATE=0 (in-process mutex only, as in round 5):"
  HIPMF_PROCESS_GATE=0 timeout 300 python tools/soak_two_processes.py 1000 1500 ) > $OUT/two_processes.txt 2>&1
cat $OUT/two_processes.txt
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python bench.py --no-configs --nrhs 0 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_small.json 2> $OUT/bench_small.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06i/bench_small.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'host', d.get('value_host_boundary_ms'))
PY
