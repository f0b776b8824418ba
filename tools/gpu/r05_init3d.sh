cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05init3d
rm -f gpurun_out/r05init3d/init.txt
for rep in 1 2 3; do
timeout 300 python tools/init_3d_lower.py 200 2>&1 | grep -E "host pieces|plan pieces|initialize wall" | cut -c1-330 >> gpurun_out/r05init3d/init.txt
done
HIPMF_PLAN_THREAD=0 timeout 300 python tools/init_3d_lower.py 200 2>&1 | grep -E "host pieces|plan pieces|initialize wall" | cut -c1-330 >> gpurun_out/r05init3d/init.txt
python tools/init_phases.py 1000 2>&1 | grep -E "initialize wall|host pieces" | cut -c1-330 >> gpurun_out/r05init3d/init.txt
python tools/init_phases.py 1000 2>&1 | grep -E "initialize wall" >> gpurun_out/r05init3d/init.txt
cat gpurun_out/r05init3d/init.txt
timeout 600 python -m pytest tests/test_fused_solve_gpu.py tests/test_round5_gpu.py tests/test_round3_gpu.py -m gpu -q -x 2>&1 | tail -2
