# the round-4 measurement set: every file lands under gpurun_out/r04final/ and is copied to profiles/ by hand
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04final
mkdir -p $OUT
export TMPDIR=/tmp
# 0. what kind of box is this (the same build gave 7.2 and 11.4 ms of numeric LU on different boxes of the pool)
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 value', d['value'], 'factor', d['phases_ms']['factor'], 'sptrsv', d['phases_ms']['sptrsv_pair'], 'solve', d['phases_ms']['solve_total_last'], 'copy GB/s', d['roofline']['measured_copy_gbs'])"; }
( run default; HIPMF_EA_LDS=0 run extend_add_read_modify_write; HIPMF_EA_LU=0 run first_tiles_by_the_panel_step; HIPMF_UPD_XCD=0 run plain_tile_order; HIPMF_MID_FRONT=0 run tiled_only; HIPMF_FACTOR_GRAPH=1 run graph; HIPMF_BLOCK_INV=1 run one_launch_steps; HIPMF_UPD_SPLIT=1000 HIPMF_EA_LDS=0 run split_updates_1000; HIPMF_MID_LU_SPLIT=1 run front_lu_by_lds_class; run default ) > $OUT/variants.txt 2>&1
cat $OUT/variants.txt
# 1. the driver's bench command
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
# 2. kernel stats of the headline command (--no-extras)
cd /tmp && rm -rf /tmp/prof_ks && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /tmp/prof_ks.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_ks -name '*.db' | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1
python tools/factor_sequence.py $DB > $OUT/factor_sequence.txt 2>&1
head -14 $OUT/kernel_stats.txt
# 3. HBM counters, separate passes; 4. FP64 matrix-pipe counters of the headline
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 400 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pmc_mfma -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_pmc.py $(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1) > $OUT/pmc_hbm.txt 2>&1
python tools/rocpd_pmc.py $(find /tmp/pmc_mfma -name '*.db' | head -1) > $OUT/pmc_mfma_c2.txt 2>&1
head -12 $OUT/pmc_mfma_c2.txt
# 5. per-level trace of the upper launches
timeout 200 python tools/fused_trace_run.py $OUT/trace.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py $OUT/trace.raw > $OUT/solve_trace.txt 2>&1
rm -f $OUT/trace.raw
# 6. many right-hand sides; config 4 in full on one GPU
HIPMF_BLOCK_COLS=16 timeout 300 python tools/many_rhs.py 2d 1000 64 > $OUT/many_rhs.txt 2>&1
HIPMF_BLOCK_COLS=16 timeout 300 python tools/many_rhs.py 3d 100 64 >> $OUT/many_rhs.txt 2>&1
cat $OUT/many_rhs.txt | tail -4
timeout 900 python tools/config4_one_gpu.py 200 256 > $OUT/config4_one_gpu.txt 2>&1
tail -5 $OUT/config4_one_gpu.txt
# 7. host phases of initialize; the microbenchmarks of this round
python tools/init_phases.py 1000 2>&1 | grep -v "^solver_hipmf" | tail -3 > $OUT/init_phases.txt
python tools/init_phases.py 100 3d sym 2>&1 | grep -v "^solver_hipmf" | tail -3 >> $OUT/init_phases.txt
./tools/microbench/front_bench 1 > $OUT/front_bench.txt 2>&1
./tools/microbench/tile_bench > $OUT/tile_bench.txt 2>&1
# 8. config 5 (Radau5 + Brusselator, npoint 513) end to end
( time timeout 600 ./russell_amd/lib/brusselator_pde --npoint 513 -g hipmf ) > $OUT/config5_radau5_brusselator_513.txt 2>&1
tail -12 $OUT/config5_radau5_brusselator_513.txt
