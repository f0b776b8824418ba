cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06j
mkdir -p $OUT
for pg in 1 0 1 0; do
  HIPMF_PROCESS_GATE=$pg timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/b_$pg.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06j/b_$pg.json').read().strip().split('\n')[-1])
print('process gate $pg: value', d['value'], 'factor', d['phases_ms']['factor'], 'pair', d['phases_ms']['sptrsv_pair'], 'frac', d['roofline']['frac'])
PY
done
