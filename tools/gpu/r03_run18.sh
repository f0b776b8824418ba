cd $GRAFT_REPO_ROOT
for t in 16 32 64 128; do HIPMF_ND_THREADS=$t python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from russell_amd import problems as P
from russell_amd.backend import Hipmf
n, rp, ci, v = P.poisson2d(1000)
best = 1e9
for rep in range(3):
    s = Hipmf(); t0 = time.perf_counter(); assert s.initialize(n, rp, ci) == 0; dt = time.perf_counter() - t0; st = s.stats(); s.close(); best = min(best, dt)
print("threads", os.environ["HIPMF_ND_THREADS"], "initialize best %.3f s ordering %.3f symbolic %.3f" % (best, st["ordering_s"], st["symbolic_s"]), flush=True)
PY
done
