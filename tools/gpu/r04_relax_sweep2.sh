cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for r in "4,16,48,0.8,0.1,0.05" "4,16,48,0.5,0.05,0.02" "4,16,48,0.8,0.05,0.02" "4,16,48,0.8,0.02,0.01" "4,16,48,0.8,0.1,0.0" "4,16,48,0.8,0.0,0.0" "4,16,48,0.5,0.0,0.0" "4,16,48,0.3,0.05,0.02" "4,8,48,0.5,0.05,0.02" "2,16,48,0.5,0.05,0.02" "4,16,32,0.5,0.05,0.02" "4,16,48,0.5,0.03,0.01" "4,16,48,0.5,0.07,0.03" "4,16,48,0.5,0.05,0.02" "4,16,48,0.8,0.1,0.05"; do
  HIPMF_RELAX=$r python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('relax %-24s value %.3f ms  factor %.3f  sptrsv_pair %.4f  nsuper %d levels %d nnzL %d launches %d' % ('$r', d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['factor']['nsuper'], d['factor']['nlevels'], d['factor']['nnz_l'], d['factor']['factor_launches']))"
done | tee gpurun_out/r04s/relax_sweep2.txt
