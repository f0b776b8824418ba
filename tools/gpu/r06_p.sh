# round 6: k_wt_bwd re-arms the tagged words (HIPMF_REARM_TAGS), off / on alternating in one call
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06p
mkdir -p $OUT
for ra in 0 1 0 1; do
  HIPMF_REARM_TAGS=$ra timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/b_$ra.json 2>/dev/null
  python - <<PY >> $OUT/rearm.txt
import json
d=json.loads(open('gpurun_out/r06p/b_$ra.json').read().strip().split('\n')[-1])
print('HIPMF_REARM_TAGS=$ra: value', d['value'], 'factor', d['phases_ms']['factor'], 'pair', d['phases_ms']['sptrsv_pair'], 'frac', d['roofline']['frac'], 'solve_total_last', d['phases_ms']['solve_total_last'], 'rel err', d['relative_error'])
PY
done
cat $OUT/rearm.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "fused or tree or parity or round5 or round6 or soak" 2>&1 | tail -3
timeout 300 python tools/soak.py 2>&1 | tail -3
