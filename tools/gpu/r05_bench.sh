cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05bench
( time timeout 1200 python bench.py ) > gpurun_out/r05bench/bench.json 2> gpurun_out/r05bench/bench.err
tail -3 gpurun_out/r05bench/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05bench/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'])
print('cpu', d['cpu_baseline']['value'], d['speedup_repeat_call'], d['speedup_one_shot'], 'total_ifs', d['total_ifs_ms'], 'host', d['value_host_boundary_ms'], d['host_api']['first_two_solves_ms'])
print('config3', {k:(v['initialize_s'],v['factorize_ms'],v['solve_ms'],v['perturbed_pivots'],v['relative_error']) for k,v in d['config3'].items() if isinstance(v,dict)})
print('config4', d['config4']['solve_s'], d['config4'].get('solve_repeat_s'), 'config5', d['config5']['ms_total'])
print('many', d['many_rhs']['solve_ms'])
PY
