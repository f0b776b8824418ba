cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06n
mkdir -p $OUT
for i in 1 2 3; do python tools/init_phases.py 1000 2>&1 | grep -v "^solver_hipmf" | tail -3; done > $OUT/init_phases.txt
cat $OUT/init_phases.txt
python tools/init_3d_lower.py 2>&1 | tail -4
timeout 900 python -m pytest tests -m gpu -q -x -k "ordering or parity or tree or round2" 2>&1 | tail -3
