cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05last
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r05last/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05last/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['phases_ms'], d['roofline']['frac'], d['value_host_boundary_ms'], d['cpu_baseline']['value'])"
