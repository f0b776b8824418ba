cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06k
mkdir -p $OUT
( echo "two processes on one GPU, 1000 x 1000 Poisson, 1500 solves each (tools/soak_two_processes.py), shared robust mutex:"
  timeout 300 python tools/soak_two_processes.py 1000 1500
  echo "the same with HIPMF_PROCESS_GATE=0 (in-process mutex only, as in round 5):"
  HIPMF_PROCESS_GATE=0 timeout 300 python tools/soak_two_processes.py 1000 1500 ) > $OUT/two_processes.txt 2>&1
cat $OUT/two_processes.txt
ls -la /dev/shm/ | head
for pg in 1 0 1 0; do
  HIPMF_PROCESS_GATE=$pg timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/b_$pg.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06k/b_$pg.json').read().strip().split('\n')[-1])
print('process gate $pg: value', d['value'], 'factor', d['phases_ms']['factor'], 'pair', d['phases_ms']['sptrsv_pair'], 'frac', d['roofline']['frac'])
PY
done
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -4
