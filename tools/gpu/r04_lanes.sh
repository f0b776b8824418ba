# many right-hand sides (64 = four blocks of 16) with one and two solve lanes, repeated: time per right-hand side and hand-off time-outs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for rep in 1 2 3 4 5 6; do for lanes in 2 1; do
  HIPMF_SOLVE_LANES=$lanes python bench.py --steps 3 --warmup 1 --no-cpu-baseline --grid3d 0 --nrhs 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
m=d['many_rhs']
print('lanes $lanes rep $rep: %.3f ms per rhs (solve %.1f ms), fallbacks %s' % (m['solve_ms']/m['nrhs_total'], m['solve_ms'], m.get('fused_fallbacks', m.get('fallbacks'))))"
done; done | tee gpurun_out/r04s/lanes.txt
