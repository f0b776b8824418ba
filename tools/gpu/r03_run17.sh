cd $GRAFT_REPO_ROOT
cat > /tmp/it.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from russell_amd import problems as P
from russell_amd.backend import Hipmf
for kind, size in (("2d", 1000), ("3d", 100)):
    n, rp, ci, v = P.poisson2d(size) if kind == "2d" else P.poisson3d(size)
    if kind == "3d":
        rp, ci, v = P.lower_triangle(n, rp, ci, v)
    for rep in range(2):
        s = Hipmf(); t0 = time.perf_counter()
        assert s.initialize(n, rp, ci, verbose=True, general_symmetric=(kind == "3d")) == 0
        print(kind, size, "initialize wall %.3f s" % (time.perf_counter() - t0), flush=True); s.close()
PY
python /tmp/it.py 2>&1 | grep -v "^solver_hipmf"
nproc
