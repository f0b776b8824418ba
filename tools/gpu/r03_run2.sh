set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 200 python tools/wt_stamps.py tools/ab/librussell_hipmf_stamps.so 1000 > gpurun_out/r03b/wt_stamps.txt 2>&1
tail -75 gpurun_out/r03b/wt_stamps.txt
for st in 0 16; do
HIPMF_UP_STAGE=$st timeout 200 python tools/fused_trace_run.py gpurun_out/r03b/trace_stage$st.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py gpurun_out/r03b/trace_stage$st.raw > gpurun_out/r03b/trace_stage$st.txt 2>&1
cat gpurun_out/r03b/trace_stage$st.txt
done
HIPMF_TREE_SOLVE=0 timeout 200 python tools/fused_trace_run.py gpurun_out/r03b/trace_old.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py gpurun_out/r03b/trace_old.raw > gpurun_out/r03b/trace_old.txt 2>&1
cat gpurun_out/r03b/trace_old.txt
rm -f gpurun_out/r03b/*.raw
