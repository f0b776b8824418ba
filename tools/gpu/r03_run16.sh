cd $GRAFT_REPO_ROOT
timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
