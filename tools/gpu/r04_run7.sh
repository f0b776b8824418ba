# eager launches against graph replay of the levels, twice (boxes differ: on a slow host the eager stream may be host-bound)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)|MHz" | head -4
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'solve', d['phases_ms']['solve_total_last'], 'copy', d['roofline']['measured_copy_gbs'])"; }
for i in 1 2; do
HIPMF_FACTOR_GRAPH=0 run eager
HIPMF_FACTOR_GRAPH=1 run graph
done
