# blocked solves under the microscope: per-kernel stats of config 4's shard (200^3 x 32 random columns), per-level stamps of a 16-column
# block at C2 and at 100^3, config 4 in full with independent random right-hand sides
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_c4 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o run -- python $GRAFT_REPO_ROOT/tools/config4_one_gpu.py 200 32 > $GRAFT_REPO_ROOT/$OUT/config4_shard.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_c4 -name '*.db' | head -1) > $OUT/config4_shard_kernel_stats.txt 2>&1
head -24 $OUT/config4_shard_kernel_stats.txt | cut -c1-150
tail -1 $OUT/config4_shard.txt | cut -c1-600
timeout 200 python tools/fused_trace_run.py $OUT/trace16.raw 1000 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/trace16.raw > $OUT/solve_trace_c2_16col.txt 2>&1
rm -f $OUT/trace16.raw
TRACE_3D=1 timeout 300 python tools/fused_trace_run.py $OUT/trace3d.raw 100 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/trace3d.raw > $OUT/solve_trace_100cube_16col.txt 2>&1
rm -f $OUT/trace3d.raw
tail -30 $OUT/solve_trace_100cube_16col.txt | cut -c1-200
timeout 900 python tools/config4_one_gpu.py 200 256 > $OUT/config4_one_gpu.txt 2>&1
tail -1 $OUT/config4_one_gpu.txt | cut -c1-700
