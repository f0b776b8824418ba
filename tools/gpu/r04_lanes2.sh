cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for rep in 1 2 3; do for lanes in 2 1; do
  HIPMF_SOLVE_LANES=$lanes python bench.py --steps 3 --warmup 1 --no-cpu-baseline --grid3d 0 --nrhs 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
m=d['many_rhs']
print('256 rhs, lanes $lanes rep $rep: %.3f ms per rhs (solve %.1f ms)' % (m['solve_ms']/m['nrhs_total'], m['solve_ms']))"
done; done | tee gpurun_out/r04s/lanes256.txt
HIPMF_SOLVE_LANES=1 python tools/many_rhs.py 2>&1 | tail -3 | tee -a gpurun_out/r04s/lanes256.txt
