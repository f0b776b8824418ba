# initialize phases on the device: 1000^2, 100^3 and 200^3 (symmetric lower) with the verbose print-out
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
{ python tools/init_phases.py 1000; python tools/init_phases.py 100 3d sym; timeout 600 python tools/init_phases.py 200 3d sym; } 2>&1 | grep -v "^solver_hipmf" | tee gpurun_out/r04h/init_phases_${TAG:-new}.txt
