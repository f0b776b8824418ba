# the round-3 measurement set: every file lands under gpurun_out/r03final/ and is copied to profiles/ by hand
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03final
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the driver's bench command
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
# 2. kernel stats of the headline command (--no-extras)
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
head -14 $OUT/kernel_stats.txt
# 3. HBM counters, separate passes
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > $OUT/pmc_hbm.txt 2>&1
# 4. per-level trace of the upper launches, wave-subtree stamps, schedule variants
timeout 200 python tools/fused_trace_run.py $OUT/trace.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py $OUT/trace.raw > $OUT/solve_trace.txt 2>&1
rm -f $OUT/trace.raw
timeout 300 python tools/solve_variants.py 1000 > $OUT/solve_variants_c2.txt 2>&1
timeout 300 python tools/solve_variants.py 100 3d "only=round-2" > $OUT/solve_variants_3d100.txt 2>&1
timeout 300 python tools/solve_variants.py 100 3d "only=tree (defaults)" >> $OUT/solve_variants_3d100.txt 2>&1
# 5. many right-hand sides
for bc in 8 16; do
echo "== HIPMF_BLOCK_COLS=$bc" >> $OUT/many_rhs.txt
HIPMF_BLOCK_COLS=$bc timeout 300 python tools/many_rhs.py 2d 1000 64 >> $OUT/many_rhs.txt 2>&1
HIPMF_BLOCK_COLS=$bc timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/many_rhs.txt 2>&1
done
cat $OUT/many_rhs.txt
# 6. config 4 in full on one GPU
timeout 900 python tools/config4_one_gpu.py 200 256 > $OUT/config4_one_gpu.txt 2>&1
tail -5 $OUT/config4_one_gpu.txt
# 7. MFMA ceiling
./tools/microbench/mfma_peak > $OUT/mfma_ceiling.txt 2>&1
# 8. host phases of initialize, chained tiled steps against one launch per step
python tools/init_phases.py 1000 2>&1 | grep -v "^solver_hipmf" | tail -3 > $OUT/init_phases.txt
python tools/init_phases.py 100 3d sym 2>&1 | grep -v "^solver_hipmf" | tail -3 >> $OUT/init_phases.txt
timeout 600 python tools/chain_check.py > $OUT/chain_check.txt 2>&1
tail -4 $OUT/chain_check.txt
