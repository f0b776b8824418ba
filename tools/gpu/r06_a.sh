# round 6, first call: the GPU suite on the block-group build + groups 1 / 2 / 4 on the many-RHS workloads
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06a
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 600 python tools/block_groups.py 2d 1000 256 1 2 4 > $OUT/block_groups.txt 2>&1
timeout 600 python tools/block_groups.py 3d 100 64 1 2 4 >> $OUT/block_groups.txt 2>&1
timeout 600 python tools/block_groups.py 3dl 100 64 1 2 4 >> $OUT/block_groups.txt 2>&1
timeout 600 python tools/block_groups.py 2d 2000 64 1 4 >> $OUT/block_groups.txt 2>&1
cat $OUT/block_groups.txt
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'])
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('speedup_repeat_call'), d.get('speedup_one_shot'), 'total_ifs', d.get('total_ifs_ms'), 'host', d.get('value_host_boundary_ms'))
print('tier2', d.get('cpu_baseline', {}).get('tier2_superlu'))
print('many', d['many_rhs']['solve_ms'], d['many_rhs']['roofline'], d['many_rhs'].get('multi_gpu_model'))
print('config4', d.get('config4'))
print('config5', {k: d['config5'].get(k) for k in ('ms_total','ms_factor_max','ms_lin_sol_max','fused_fallbacks','gate_waits')} if 'config5' in d else None)
PY
