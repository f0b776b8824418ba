cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05acc
timeout 400 python tools/omega_probe.py 2>&1 | head -3 > gpurun_out/r05acc/o.txt
timeout 300 python tools/many_rhs.py 2d 1000 64 >> gpurun_out/r05acc/o.txt 2>&1
timeout 300 python tools/many_rhs.py 3d 100 64 >> gpurun_out/r05acc/o.txt 2>&1
timeout 900 python tools/config4_one_gpu.py 200 32 2>&1 | grep -o '"solve_all_ms[^}]*' | cut -c1-330 >> gpurun_out/r05acc/o.txt
timeout 900 python tools/config4_one_gpu.py 200 32 2>&1 | grep -o '"solve_all_ms[^,]*' >> gpurun_out/r05acc/o.txt
cat gpurun_out/r05acc/o.txt
