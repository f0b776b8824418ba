cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04j
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py -m gpu -q 2>&1 | tail -8
for i in 1 2; do
HIPMF_BLOCK_INV=0 run binv_off
HIPMF_BLOCK_INV=1 run binv_on
done 2>&1 | tee gpurun_out/r04j/binv_ab.txt
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04j/kernel_stats_binv.txt 2>&1
python tools/factor_sequence.py $DB > gpurun_out/r04j/factor_sequence_binv.txt 2>&1
head -12 gpurun_out/r04j/kernel_stats_binv.txt
