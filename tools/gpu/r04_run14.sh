cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'assemble', d['phases_ms']['assemble'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_round3_gpu.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2 3; do
HIPMF_EA_LDS=0 run ea_rmw
HIPMF_EA_LDS=1 run ea_lds
done 2>&1 | tee gpurun_out/r04k/ea_ab.txt
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04k/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > gpurun_out/r04k/factor_sequence.txt 2>&1
head -12 gpurun_out/r04k/kernel_stats.txt
