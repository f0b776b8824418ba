cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05last2
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r05last2/pytest_gpu.txt 2>&1
grep -E "passed|failed" gpurun_out/r05last2/pytest_gpu.txt | tail -2
( time timeout 1200 python bench.py ) > gpurun_out/r05last2/bench.json 2> gpurun_out/r05last2/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05last2/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'])
print('cpu', d['cpu_baseline']['value'], d['speedup_repeat_call'], d['speedup_one_shot'], 'total_ifs', d['total_ifs_ms'], 'host', d['value_host_boundary_ms'])
print('many', d['many_rhs']['solve_ms'], d['many_rhs']['roofline'])
print('config4', d['config4']['solve_s'], d['config4'].get('solve_repeat_s'), d['config4']['max_relative_error_all_columns'], 'config5', d['config5']['ms_total'])
print('poisson3d', d['poisson3d'].get('sptrsv_frac_of_hbm_peak'))
PY
timeout 900 python tools/config4_one_gpu.py 200 256 > gpurun_out/r05last2/config4_one_gpu.txt 2>&1
tail -1 gpurun_out/r05last2/config4_one_gpu.txt | cut -c1-500
timeout 300 python tools/many_rhs.py 2d 1000 64 > gpurun_out/r05last2/many_rhs.txt 2>&1
timeout 300 python tools/many_rhs.py 3d 100 64 >> gpurun_out/r05last2/many_rhs.txt 2>&1
cat gpurun_out/r05last2/many_rhs.txt
