cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05bfs
for v in 2000000000 1500000 2000000000 1500000; do
echo "== HIPMF_ND_PAR_BFS=$v" >> gpurun_out/r05bfs/init.txt
HIPMF_ND_PAR_BFS=$v timeout 300 python tools/init_3d_lower.py 200 2>&1 | grep -E "ordering|initialize wall" | cut -c1-260 >> gpurun_out/r05bfs/init.txt
done
cat gpurun_out/r05bfs/init.txt
