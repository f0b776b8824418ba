cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05am
for t in 16 32 64 8; do
echo "== HIPMF_ND_THREADS=$t" >> gpurun_out/r05am/init.txt
for rep in 1 2; do
HIPMF_ND_THREADS=$t python tools/init_phases.py 1000 2>&1 | grep -E "ordering|initialize wall" | sed 's/; plan.*//' >> gpurun_out/r05am/init.txt
done
done
cat gpurun_out/r05am/init.txt
