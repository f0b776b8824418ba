cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" HIPMF_UP_REPLICAS=0 > gpurun_out/r03f/solve_variants_c2.txt 2>&1
timeout 300 python tools/solve_variants.py 1000 "only=round-2" HIPMF_UP_STAGE_MID=0 >> gpurun_out/r03f/solve_variants_c2.txt 2>&1
cat gpurun_out/r03f/solve_variants_c2.txt
timeout 200 python tools/fused_trace_run.py gpurun_out/r03f/trace.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py gpurun_out/r03f/trace.raw > gpurun_out/r03f/trace_default.txt 2>&1
cat gpurun_out/r03f/trace_default.txt
rm -f gpurun_out/r03f/*.raw
timeout 900 python -m pytest tests/test_fused_solve_gpu.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
