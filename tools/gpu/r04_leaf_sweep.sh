# nested-dissection leaf size against the headline (the kernels of round 4 under other trees): value / factor / SpTRSV pair per setting
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
for leaf in 16 8 12 24 32 48; do
  HIPMF_ND_LEAF=$leaf python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('leaf $leaf: value %.3f ms  factor %.3f  sptrsv_pair %.4f  solve %.3f  nsuper %d levels %d nnzL %d init %.0f ms' % (d['value'], d['phases_ms']['factor'], d['phases_ms']['sptrsv_pair'], d['phases_ms']['solve_total_last'], d['factor']['nsuper'], d['factor']['nlevels'], d['factor']['nnz_l'], d['phases_ms']['initialize_once']))"
done | tee gpurun_out/r04s/leaf_sweep.txt
