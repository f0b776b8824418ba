cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 300 python tools/solve_variants.py 1000 > gpurun_out/r03e/solve_variants_c2.txt 2>&1
cat gpurun_out/r03e/solve_variants_c2.txt
timeout 200 python tools/fused_trace_run.py gpurun_out/r03e/trace.raw 1000 > /dev/null 2>&1
python tools/fused_trace.py gpurun_out/r03e/trace.raw > gpurun_out/r03e/trace_default.txt 2>&1
cat gpurun_out/r03e/trace_default.txt
rm -f gpurun_out/r03e/*.raw
timeout 600 python -m pytest tests/test_fused_solve_gpu.py -x -q 2>&1 | tail -3
timeout 300 python tools/solve_variants.py 100 3d "only=tree (defaults)" 2>&1 | tail -2
timeout 300 python tools/solve_variants.py 100 3d "only=round-2" 2>&1 | tail -1
