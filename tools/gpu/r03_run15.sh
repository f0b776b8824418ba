cd $GRAFT_REPO_ROOT
for tf in 16 40 70 120 200; do
echo "top fronts $tf"; HIPMF_UP_TOP_FRONTS=$tf timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
done
for st in "24 32" "32 32" "32 64" "40 48"; do
set -- $st
echo "stage $1 / $2"; HIPMF_UP_STAGE=$1 HIPMF_UP_STAGE_BWD=$2 timeout 300 python tools/solve_variants.py 1000 "only=tree (defaults)" 2>&1 | tail -1
done
