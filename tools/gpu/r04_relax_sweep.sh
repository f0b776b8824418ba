# relaxed amalgamation against the headline (HIPMF_RELAX = n0,n1,n2,z0,z1,z2; default 4,16,48,0.8,0.1,0.05)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for r in "4,16,48,0.8,0.1,0.05" "0,0,0,0,0,0" "8,16,48,0.8,0.1,0.05" "4,32,64,0.8,0.2,0.1" "16,32,64,0.8,0.3,0.1" "4,16,48,0.8,0.3,0.2" "32,48,64,0.9,0.5,0.3" "4,16,48,0.5,0.05,0.02"; do
  HIPMF_RELAX=$r python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('relax %-24s value %.3f ms  factor %.3f  sptrsv_pair %.4f  nsuper %d levels %d nnzL %d launches %d' % ('$r', d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['factor']['nsuper'], d['factor']['nlevels'], d['factor']['nnz_l'], d['factor']['factor_launches']))"
done | tee gpurun_out/r04s/relax_sweep.txt
