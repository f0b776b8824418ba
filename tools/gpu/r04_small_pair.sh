cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for rep in 1 2 3; do for sp in 0 1; do
  HIPMF_SMALL_PAIR=$sp python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('small_pair $sp rep $rep: value %.3f ms  factor %.3f  rel err %.1e' % (d['value'], d['phases_ms']['factor'], d['relative_error']))"
done; done | tee gpurun_out/r04s/small_pair.txt
