# initialize after the subtree-parallel elimination tree / column counts: 1000^2 and 200^3 (symmetric lower), verbose print-out
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
export TMPDIR=/tmp
{ INIT_REPS=4 python tools/init_phases.py 1000; INIT_REPS=2 timeout 600 python tools/init_phases.py 200 3d sym; } 2>&1 | grep -v "^solver_hipmf" > gpurun_out/r04j/init_phases.txt
grep -E "graph .* ordering|wall" gpurun_out/r04j/init_phases.txt | cut -c1-260
