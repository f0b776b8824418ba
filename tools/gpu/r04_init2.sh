# initialize after the threaded extend-add plan: 1000^2 and 200^3 (symmetric lower) with the verbose print-out, then a short bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
{ python tools/init_phases.py 1000; INIT_REPS=2 timeout 600 python tools/init_phases.py 200 3d sym; } 2>&1 | grep -v "^solver_hipmf" | tee gpurun_out/r04i/init_phases.txt
( time timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i/bench.json').read().strip().split('\n')[0])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'total_ifs', d.get('total_ifs_ms'))
PY
