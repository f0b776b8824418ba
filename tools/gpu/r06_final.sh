# the round-6 measurement set: every file lands under gpurun_out/r06final/ and is copied to profiles/ by hand
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06final
mkdir -p $OUT
export TMPDIR=/tmp
# 0. the GPU suite and smoke on this build
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
# 1. the driver's bench command (defaults)
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06final/bench.json').read().strip().split('\n')[-1])
print('value', d['value'], d['phases_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('speedup_repeat_call'), d.get('speedup_one_shot'), 'total_ifs', d.get('total_ifs_ms'), 'host', d.get('value_host_boundary_ms'))
print('many', d['many_rhs']['solve_ms'], d['many_rhs']['roofline'], d['many_rhs'].get('multi_gpu_model'))
print('config4', d.get('config4'))
print('tier2', d.get('cpu_baseline', {}).get('tier2_superlu'))
print('config5', {k: d['config5'].get(k) for k in ('ms_total','ms_factor_max','ms_lin_sol_max','fused_fallbacks','gate_waits')} if 'config5' in d else None)
PY
# 2. A/B of the solve schedules in one call
timeout 300 python tools/solve_variants.py 1000 HIPMF_UP_MAX_GROUPS=16 > $OUT/solve_variants_c2.txt 2>&1
cat $OUT/solve_variants_c2.txt
# 3. kernel stats of the headline command
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
head -16 $OUT/kernel_stats.txt
# 4. HBM counters (separate passes) and FP64 matrix-pipe counters of the headline
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_small.json 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmc_mfma -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > $OUT/pmc_hbm.txt 2>&1
python tools/sptrsv_traffic.py $OUT/pmc_hbm.txt 2 1065545568 715697040 "profiles/r06_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, final build of round 6)" $OUT/bench_small.json > $OUT/sptrsv_traffic.json 2>&1
python tools/rocpd_pmc.py $(find /tmp/pmc_mfma -name '*.db' | head -1) > $OUT/pmc_mfma_c2.txt 2>&1
head -8 $OUT/pmc_mfma_c2.txt
grep ratio $OUT/sptrsv_traffic.json
# 5. per-level trace of the upper launches
timeout 200 python tools/fused_trace_run.py $OUT/trace.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py $OUT/trace.raw > $OUT/solve_trace.txt 2>&1
rm -f $OUT/trace.raw
# 6. many right-hand sides (per-kernel stats of blocked solves); config 4 in full on one GPU
cd /tmp && rm -rf /tmp/prof_many && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python $GRAFT_REPO_ROOT/tools/block_groups.py 2d 1000 256 4 > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_many -name '*.db' | head -1) > $OUT/many_rhs_kernel_stats.txt 2>&1
timeout 300 python tools/block_groups.py 2d 1000 256 1 4 > $OUT/block_groups.txt 2>&1
timeout 300 python tools/block_groups.py 3d 100 64 1 4 >> $OUT/block_groups.txt 2>&1
cat $OUT/block_groups.txt
timeout 300 python tools/many_rhs.py 2d 1000 64 > $OUT/many_rhs.txt 2>&1
timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/many_rhs.txt 2>&1
tail -2 $OUT/many_rhs.txt
timeout 900 python tools/config4_one_gpu.py 200 256 > $OUT/config4_one_gpu.txt 2>&1
tail -5 $OUT/config4_one_gpu.txt
# 7. host phases of initialize
python tools/init_phases.py 1000 2>&1 | grep -v "^solver_hipmf" | tail -3 > $OUT/init_phases.txt
cat $OUT/init_phases.txt
# 8. config 5 (Radau5 + Brusselator, npoint 513) end to end
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) > $OUT/config5_radau5_brusselator_513.txt 2>&1
tail -12 $OUT/config5_radau5_brusselator_513.txt
