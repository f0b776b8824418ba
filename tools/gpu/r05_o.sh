# per-level stamps of ONE 16-column block: C2, 100^3, 144^3 (unsymmetric storage: LU fronts)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05o
mkdir -p $OUT
timeout 200 python tools/fused_trace_run.py $OUT/t.raw 1000 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/t.raw > $OUT/solve_trace_c2_16col.txt 2>&1
TRACE_3D=1 timeout 300 python tools/fused_trace_run.py $OUT/t.raw 100 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/t.raw > $OUT/solve_trace_100cube_16col.txt 2>&1
TRACE_3D=1 timeout 600 python tools/fused_trace_run.py $OUT/t.raw 144 16 > /dev/null 2>&1
python tools/fused_trace.py $OUT/t.raw > $OUT/solve_trace_144cube_16col.txt 2>&1
rm -f $OUT/t.raw
cat $OUT/solve_trace_144cube_16col.txt | cut -c1-190
