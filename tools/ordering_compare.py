"""Nested dissection / approximate minimum degree / Ordering::Best on the device: initialize, factorize and solve times, fill and flops.
python tools/ordering_compare.py [circuit n | grid N | grid3d N] ..."""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402
from test_ordering_amd_gpu import _circuit_like  # noqa: E402

NAMES = {0: "nested dissection", 3: "minimum degree", 4: "best"}


def run(label, n, rp, ci, v):
    xs = P.manufactured_solution(n)
    b = sp.csr_matrix((v, ci, rp), shape=(n, n)) @ xs
    for o in (0, 3, 4):
        s = Hipmf()
        t0 = time.time()
        code = s.initialize(n, rp, ci, ordering=o)
        t1 = time.time()
        if code != 0:
            print("%s | %-17s refused (%d): %s" % (label, NAMES[o], code, (s.lib.solver_hipmf_last_error(s.h) or b"").decode()))
            s.close()
            continue
        assert s.factorize(v) == 0  # (first call: includes one-time set-up)
        t2 = time.time()
        assert s.factorize(v) == 0
        t3 = time.time()
        x = s.solve(b)
        t4 = time.time()
        st = s.stats()
        s.close()
        print("%s | %-17s initialize %.3f s, factorize %.1f ms (first %.1f), solve %.2f ms | nnz(L) %d, flops %.3e, levels %d, largest front %d, pool %.2f GB | error %.1e"
              % (label, NAMES[o], t1 - t0, 1e3 * (t3 - t2), 1e3 * (t2 - t1), 1e3 * (t4 - t3), st["nnz_l"], st["flops"], st["nlevels"], st["max_front"], st["pool_bytes"] / 1e9,
                 np.max(np.abs(x - xs)) / np.max(np.abs(xs))), flush=True)


args = sys.argv[1:] or ["circuit", "60000", "grid", "1000"]
for kind, size in zip(args[0::2], args[1::2]):
    size = int(size)
    if kind == "circuit":
        A = _circuit_like(size, 11)
        run("circuit-like n = %d" % size, size, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
    elif kind == "grid":
        run("5-point grid %d^2" % size, *P.poisson2d(size))
    else:
        run("7-point grid %d^3" % size, *P.poisson3d(size))
