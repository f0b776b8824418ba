cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for r in "4,16,64,0.8,0.1,0.05" "4,24,64,0.8,0.1,0.05" "4,32,64,0.8,0.1,0.05" "4,16,64,0.6,0.1,0.05" "4,16,64,0.8,0.12,0.05" "4,16,64,0.8,0.08,0.05" "4,16,56,0.8,0.1,0.05" "4,16,72,0.8,0.1,0.05" "6,16,64,0.8,0.1,0.05" "4,16,64,0.8,0.1,0.05"; do
  HIPMF_RELAX=$r HIPMF_RELAX_BIG=2048 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('relax %-22s big 2048  value %.3f factor %.3f pair %.4f launches %d nsuper %d' % ('$r', d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['factor']['factor_launches'], d['factor']['nsuper']))"
done | tee gpurun_out/r04s/relax_sweep5.txt
