# per-kernel totals of three blocked solves of 64 right-hand sides (1000 x 1000, no refinement): profiles/r03_many_rhs_kernel_stats.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/many_prof.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from russell_amd import problems as P
from russell_amd.backend import Hipmf
n, rp, ci, v = P.poisson2d(1000)
s = Hipmf()
assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
assert s.factorize(v) == 0
B = np.random.default_rng(0).standard_normal((64, n))
for _ in range(3):
    X = s.solve_many(B)
s.close()
PY
cd /tmp && rm -rf /tmp/prof_many && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_many -o run -- python /tmp/many_prof.py > /tmp/prof_many.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03final
DB=$(find /tmp/prof_many -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB > gpurun_out/r03final/many_rhs_kernel_stats.txt 2>&1; head -12 gpurun_out/r03final/many_rhs_kernel_stats.txt; else tail -5 /tmp/prof_many.log; fi
