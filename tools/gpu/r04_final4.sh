# final build of round 4 (amalgamation gated by the front size, one solve lane): the driver's round-end sequence, then the rocprofv3 summaries of the headline
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f6
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04f6/bench.json').read().strip().split('\n')[0])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d.get('speedup_repeat_call'), 'total_ifs', d.get('total_ifs_ms'), 'many', d['many_rhs']['solve_ms'])
PY
cd /tmp && rm -rf /tmp/prof_ks && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
head -12 $OUT/kernel_stats.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > $OUT/pmc_hbm.txt 2>&1
head -8 $OUT/pmc_hbm.txt
