cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05u
mkdir -p $OUT
TRACE_3D=1 TRACE_SYM=1 timeout 900 python tools/fused_trace_run.py /tmp/t.raw 200 16 > $OUT/run.log 2>&1
python tools/fused_trace.py /tmp/t.raw > $OUT/solve_trace_200cube_sym_16col.txt 2>&1
ls -la /tmp/t.raw; rm -f /tmp/t.raw
cut -c1-170 $OUT/solve_trace_200cube_sym_16col.txt
