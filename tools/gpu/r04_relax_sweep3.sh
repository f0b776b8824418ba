cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for r in "4,16,48,0.8,0.1,0.05" "4,16,48,0.8,0.1,0.0" "4,16,48,0.8,0.1,0.01" "4,16,48,0.8,0.07,0.0" "4,16,64,0.8,0.1,0.0" "4,16,32,0.8,0.1,0.0" "4,16,48,0.8,0.1,0.0" "4,16,48,0.8,0.1,0.05"; do
  HIPMF_RELAX=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --nrhs 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('relax %-22s value %.3f factor %.3f pair %.4f | sym %.3f / %.3f | 100^3 factor %.1f pair %.2f pool %.2f | many %.3f ms/rhs | launches %d' % ('$r', d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['symmetric']['value_ms'], d['symmetric']['factor_ms'], d['poisson3d']['factor_ms'], d['poisson3d']['sptrsv_pair_ms'], d['poisson3d'].get('pool_gb', 0), d['many_rhs']['solve_ms'] / d['many_rhs']['nrhs_total'], d['factor']['factor_launches']))"
done | tee gpurun_out/r04s/relax_sweep3.txt
