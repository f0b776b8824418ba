import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from russell_amd import problems as P
from russell_amd.backend import Hipmf
import scipy.sparse as sp
lib = sys.argv[1] if len(sys.argv)>1 and sys.argv[1] != "gpu" else None
def lower(n, rp, ci, v):
    A = sp.csr_matrix((v, ci, rp), shape=(n,n))
    L = sp.tril(A).tocsr(); L.sort_indices()
    return L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64)
cases = [("2d",(120,110)),("3d",(14,13,12)),("3d",(20,20,20))] if lib else [("2d",(1000,1000)),("3d",(64,64,64)),("3d",(100,100,100))]
for kind,dims in cases:
    n, rp, ci, v = (P.poisson2d(*dims) if kind=="2d" else P.poisson3d(*dims))
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    lrp, lci, lv = lower(n, rp, ci, v)
    for mode in ("ldlt","lu_mirrored","lu_full"):
        os.environ["HIPMF_SYM_LDLT"]="0" if mode=="lu_mirrored" else "1"
        s = Hipmf(lib)
        t0=time.time()
        if mode=="lu_full": code = s.initialize(n, rp, ci, refinement_nstep=0)
        else: code = s.initialize(n, lrp, lci, general_symmetric=True, refinement_nstep=0)
        assert code==0, code
        t1=time.time()
        code = s.factorize(v if mode=="lu_full" else lv, compute_determinant=True)
        x = s.solve(b)
        st = s.stats()
        print(kind,dims,mode,"code",code,"err %.2e"%np.max(np.abs(x-xs)),"maxfront",st["max_front"],"maxp",st["max_pivots"],"pool %.3f GB"%(st["pool_bytes"]/1e9),"det",s.det_coefficient,s.det_exponent,"factor_ms %.3f fwd %.3f bwd %.3f init %.2fs"%(st["factor_ms"],st["fwd_ms"],st["bwd_ms"],t1-t0), flush=True)
        s.close()
