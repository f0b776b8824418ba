cd $GRAFT_REPO_ROOT
timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
HIPMF_UP_STAGE_MID=0 timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
HIPMF_UP_STAGE_MID=16 timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
timeout 300 python tools/solve_variants.py 100 3d "only=tree (defaults)" 2>&1 | tail -1
